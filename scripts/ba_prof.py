#!/usr/bin/env python3
"""Phase breakdown of the local-map solver (ba_solve_dev, run by k_ba_worker) (needs a build with FLVIS_EXTRA_HIPCC_FLAGS=-DFLVIS_BA_PROF)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import flvis_amd
from flvis_amd import synth

S, N = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 110
dev = torch.device("cuda", 0)
yp = os.path.join(tempfile.gettempdir(), "baprof.yaml")
open(yp, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(yp)
ctx = flvis_amd.Context(0)
trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715)
trajs = [synth.Trajectory(s) for s in range(S)]
rnd = synth.Renderer(dev)
skip = cfg.skip_first_n_imgs
fr0 = rnd.stereo_frame(trajs, skip / synth.FRAME_HZ, skip)
lib = ctx._lib
tprev = -1.0 / synth.FRAME_HZ
for f in range(N):
    t = f / synth.FRAME_HZ
    imu = np.zeros((S, 16, 7)); cnt = np.zeros(S, np.int32)
    for i in range(S):
        smp = synth.imu_samples(trajs[i], i, tprev, t)
        imu[i, :len(smp)] = smp; cnt[i] = len(smp)
    tprev = t
    lib.flvis_imu_feed_all(ctx._h, cnt.ctypes.data_as(C.POINTER(C.c_int)), imu.ctypes.data_as(C.POINTER(C.c_double)), 16)
    i0, i1 = fr0 if f < skip else rnd.stereo_frame(trajs, t, f)
    tt = np.full(S, t)
    rc = lib.flvis_image_feed(ctx._h, C.c_void_p(i0.data_ptr()), C.c_void_p(i1.data_ptr()),
                              tt.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(0), 1)
    assert rc == 0
c = (C.c_int64 * 64)()
lib.flvis_debug_counters(ctx._h, c)
c = np.array(c[:])
names = ["misc", "structure", "schur:stage+barrier", "linearize_lm", "linearize_pose+chi2", "chol_factor", "tri_solves", "schur", "trial_poses",
         "update", "chi2_trial", "cull/finish"]
runs, trials = max(c[8 + 15], 1), max(c[8 + 14], 1)
print("ba runs %d  trials %d (%.1f/run)  edges/run %.0f  landmarks/run %.0f" % (runs, trials, trials / runs, c[8 + 16] / runs, c[8 + 17] / runs))
print("schur:accumulate %.1f us/run   schur:combine+write %.1f us/run  (then `schur` = waiting for the slowest wave + IMU blocks)" % (c[8 + 12] / runs / 100.0, c[8 + 13] / runs / 100.0))
tot = sum(c[8:8 + 14])
for i, n in enumerate(names):
    print("%-16s %8.1f us/run  %5.1f%%" % (n, c[8 + i] / runs / 100.0, 100.0 * c[8 + i] / max(tot, 1)))
print("total %.1f us/run (100 MHz counter assumed)" % (tot / runs / 100.0))
if c[40]:
    print("Schur accumulate per wave, us/run: " + "  ".join("%.1f" % (c[40 + w] / runs / 100.0) for w in range(8)))
if c[48]:
    print("partition by keyframe distance d (lanes per pair, landmarks per pair): " + "  ".join(
        "d%d: %.1f / %.0f" % (d, c[48 + 2 * d] / max(runs * 2, 1) / (7 - d), c[49 + 2 * d] / max(runs * 2, 1) / (7 - d)) for d in range(7)))
if c[27]:
    print("keyframe bookkeeping (ba_update_dev): %.1f us per keyframe (%d keyframes)" % (c[26] / c[27] / 100.0, c[27]))

if c[24 + 7] or c[32 + 7]:
    for base, name, ph in ((24, "ransac_f", ["load", "generate", "score", "replay", "mask+count"]),
                           (32, "ransac_pnp", ["gather", "generate", "score", "replay(+prev)", "mask+refine"])):
        n = max(c[base + 7], 1)
        print("%s: %d calls, %.2f batches/call: " % (name, c[base + 7], c[base + 6] / n) +
              "  ".join("%s=%.1fus" % (nm, c[base + i] / n / 100.0) for i, nm in enumerate(ph)))

if c[24 + 7]:
    n = max(c[24 + 7], 1)
    print("seven_point (thread 0): " + "  ".join("%s=%.1fus" % (nm, c[40 + i] / n / 100.0) for i, nm in enumerate(["fill", "sweeps", "finish (norms, sort, completion, cubic, matrices)", "-", "-", "load .. draw", "barrier"])))
    if c[56 + 6]:
        n = c[56 + 6]
        print("EPnP (wave 0's hypotheses, %d solves): " % n + "  ".join("%s=%.1fus" % (nm, c[56 + i] / n / 100.0) for i, nm in enumerate(
            ["head", "jacobi12", "pick+constraints", "betas", "centroids+abt", "pose"])))

if c[48 + 7]:
    n = max(c[48 + 7], 1)
    print("pose_lm: %d calls: " % c[48 + 7] + "  ".join("%s=%.1fus" % (nm, c[48 + i] / n / 100.0) for i, nm in enumerate(
        ["track_post|gather", "rank+stage", "optimise 1", "cull", "optimise 2"])))
