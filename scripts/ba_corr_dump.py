#!/usr/bin/env python3
"""Dumps the local map's last CorrectionInf of every stream (keyframe id, pose, landmark ids and positions, outlier ids) after N frames of
S synthetic streams to an .npz: run it with two builds of the library (FLVIS_LIB_PATH, scripts/build_variant.sh) and compare the files --
`scripts/ba_corr_dump.py cmp a.npz b.npz` prints whether they agree bit for bit.  Debug aid for solver changes that claim unchanged
arithmetic (needs an MI355X)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = sorted(a.files) == sorted(b.files) and all(a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes() for k in a.files)
    worst = max([float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()) for k in a.files if k in b.files and a[k].shape == b[k].shape and a[k].size] + [0.0])
    print("local-map corrections of %d arrays: %s (largest difference %.3g)" % (len(a.files), "bit-identical" if same else "DIFFERENT", worst))
    sys.exit(0 if same else 1)

import torch  # noqa
import flvis_amd
from flvis_amd import synth

out, S, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16, int(sys.argv[3]) if len(sys.argv) > 3 else 70
yp = os.path.join(tempfile.gettempdir(), "bacorr.yaml")
open(yp, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(yp)
ctx = flvis_amd.Context(0)
trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=N)
trajs = [synth.Trajectory(s) for s in range(S)]
rnd = synth.Renderer("cuda")
skip = cfg.skip_first_n_imgs
t_prev = -1.0 / synth.FRAME_HZ
standin = None
for f in range(N):
    t = f / synth.FRAME_HZ
    for i in range(S):
        trk.imu_feed_flvis(i, synth.imu_samples(trajs[i], i, t_prev, t))
    t_prev = t
    if f >= skip or standin is None:
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        standin = standin or (i0, i1)
    else:
        i0, i1 = standin
    trk.image_feed(i0, i1, [t] * S, with_local_map=True)   # (the frame's output is read back: the local map keeps pace, no keyframe waits)
arrs = {}
for i in range(S):
    c = trk.correction(i)
    if c is None:
        continue
    arrs["s%d_frame" % i] = np.array([c["frame_id"]])
    for k in ("pose7", "lm_id", "lm_3d", "outlier_id"):
        arrs["s%d_%s" % (i, k)] = np.asarray(c[k])
rows = np.stack([trk.trajectory(i, 0, N) for i in range(S)])
arrs["traj"] = rows
np.savez(out, **arrs)
print("wrote %s: %d streams with a correction, counters %s" % (out, sum(1 for k in arrs if k.endswith("_frame")), trk.counters()))
ctx.close()
