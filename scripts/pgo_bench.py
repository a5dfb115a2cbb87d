#!/usr/bin/env python3
"""Time of flvis_hip_pgo_loop_closure for synthetic loops of growing size (one graph, and a batch of 16 graphs of the same size)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import flvis_amd
import _pgo_synth as PS

ctx = flvis_amd.Context(0)
for n_kf in (70, 130, 260, 520):
    c = PS.make_loop(3, n_kf=n_kf, extra_loops=1)
    pres = np.ones(n_kf, np.uint8)
    for batch in (1, 16):
        args = ([c["est"]] * batch, [pres] * batch, [c["loops"]] * batch, [c["loop_poses"]] * batch)
        ctx.pgo_loop_closure(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, stats, ran = ctx.pgo_loop_closure(*args)
        dt = (time.perf_counter() - t0) * 1e3
        print("keyframes %4d  graphs %2d  %.1f ms  (%d LM iterations, chi2 %.3g -> %.3g)" % (n_kf, batch, dt, stats[0][0], stats[0][1], stats[0][2]))
