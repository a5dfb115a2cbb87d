#!/usr/bin/env python3
"""How full the chip is during the two LK launches of a frame (needs scripts/build_variant.sh lkutil lk_kernel.hip -DFLVIS_LK_UTIL and
FLVIS_LIB_PATH on that library): per launch kind the waves' summed durations against (latest end - earliest start) x wave slots, frame by
frame (one frame per call, synchronised) with and without the local map beside it.

usage: python scripts/lk_util.py [frames=140] [local_map=1]"""
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

S, N = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 140
WLM = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
yp = os.path.join(tempfile.gettempdir(), "lkutil.yaml")
open(yp, "w").write(synth.D435I_STEREO_YAML)
cfg = flvis_amd.load_config(yp)
ctx = flvis_amd.Context(0)
trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715)
trajs = [synth.Trajectory(s) for s in range(S)]
rnd = synth.Renderer(dev)
skip = cfg.skip_first_n_imgs
fr0 = rnd.stereo_frame(trajs, skip / synth.FRAME_HZ, skip)
lib = ctx._lib
lib.flvis_debug_lk_util.argtypes = [C.c_void_p, C.c_int]
NV = 3 * 8 * 32768 * 2
buf = (C.c_ulonglong * NV)()
tprev = -1.0 / synth.FRAME_HZ
# pre-render the frames so that the renderer's kernels are not beside the tracker's
frames = {}
for f in range(N):
    t = f / synth.FRAME_HZ
    frames[f] = fr0 if f < skip else rnd.stereo_frame(trajs, t, f)
torch.cuda.synchronize()
imus, cnts, tts = [], [], []
for f in range(N):   # every frame's inputs staged first: the feeding loop below must keep ahead of a 1 ms frame
    t = f / synth.FRAME_HZ
    imu = np.zeros((S, 16, 7)); cnt = np.zeros(S, np.int32)
    for i in range(S):
        smp = synth.imu_samples(trajs[i], i, tprev, t)
        imu[i, :len(smp)] = smp; cnt[i] = len(smp)
    tprev = t
    imus.append(imu); cnts.append(cnt); tts.append(np.full(S, t))
for f in range(N):
    if f == N - 8:   # the last 8 frames are measured, free-running (no readback): slot = frame number mod 8
        lib.flvis_debug_lk_util(None, 1)
    lib.flvis_imu_feed_all(ctx._h, cnts[f].ctypes.data_as(C.POINTER(C.c_int)), imus[f].ctypes.data_as(C.POINTER(C.c_double)), 16)
    i0, i1 = frames[f]
    rc = lib.flvis_image_feed(ctx._h, C.c_void_p(i0.data_ptr()), C.c_void_p(i1.data_ptr()), tts[f].ctypes.data_as(C.POINTER(C.c_double)),
                              C.c_void_p(0), WLM)
    assert rc == 0
lib.flvis_hip_synchronize(ctx._h)
lib.flvis_debug_lk_util(buf, 0)
v = np.frombuffer(buf, dtype=np.uint64).reshape(3, 8, 32768, 2).astype(np.float64) / 100.0   # us
for role, name in ((1, "temporal"), (2, "stereo")):
    spans, busys, ns, means, tails, p99s = [], [], [], [], [], []
    for k in range(8):
        w = v[role % 3][k]
        w = w[w[:, 1] > 0]
        if len(w) == 0:
            continue
        t0, t1 = w[:, 0].min(), w[:, 1].max()
        d = w[:, 1] - w[:, 0]
        spans.append(t1 - t0); busys.append(d.sum()); ns.append(len(w)); means.append(d.mean()); p99s.append(np.percentile(d, 99))
        # the tail: from the moment the last wave STARTS to the end of the launch (no wave is waiting for a slot any more)
        tails.append(t1 - w[:, 0].max())
    if not spans:
        continue
    span, busy, n = np.mean(spans), np.mean(busys), np.mean(ns)
    print("%-8s launches %d  waves %.0f  span %.1f us  wave mean %.1f / p99 %.1f us  in flight on average %.0f = %.2f of 4096 slots;  last wave starts %.1f us before the end"
          % (name, len(spans), n, span, np.mean(means), np.mean(p99s), busy / span, busy / span / 4096.0, np.mean(tails)))
