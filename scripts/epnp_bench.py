#!/usr/bin/env python3
"""time of flvis_hip_debug_epnp (one wavefront per correspondence set) for a few set sizes"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import flvis_amd
import _geom as G

K4 = np.array([435.2, 435.2, 367.4, 252.2])
ctx = flvis_amd.Context(0)
rng = np.random.default_rng(0)
for n, sets in ((5, 64), (5, 512), (200, 64)):
    cap = 256
    p3 = np.zeros((sets, cap, 3), np.float32)
    p2 = np.zeros((sets, cap, 2), np.float32)
    for k in range(sets):
        P, _ = G.random_scene(rng, n, K4)
        R = G.rodrigues(rng.normal(0, 0.2, 3))
        t = rng.normal(0, 0.3, 3)
        Pw = (P - t) @ R
        p3[k, :n] = Pw
        p2[k, :n] = G.project(R, t, Pw, K4) + rng.normal(0, 0.3, (n, 2))
    d3, d2 = torch.from_numpy(p3).cuda(), torch.from_numpy(p2).cuda()
    cnt = torch.full((sets,), n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.debug_epnp(d3, d2, cnt, K4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ctx.debug_epnp(d3, d2, cnt, K4)
    e1.record()
    torch.cuda.synchronize()
    print("n = %d, %d sets: %.1f us per launch" % (n, sets, e0.elapsed_time(e1) / 20 * 1e3))

# the whole PnP RANSAC of the loop closing's geometric check (SOLVEPNP_P3P: P3P hypotheses, EPnP on the inliers) on 64 sets
for n in (120, 400):
    sets, cap = 64, 512
    p3 = np.zeros((sets, cap, 3), np.float32)
    p2 = np.zeros((sets, cap, 2), np.float32)
    for k in range(sets):
        P, _ = G.random_scene(rng, n, K4)
        R = G.rodrigues(rng.normal(0, 0.2, 3))
        t = rng.normal(0, 0.3, 3)
        Pw = (P - t) @ R
        z = G.project(R, t, Pw, K4) + rng.normal(0, 0.4, (n, 2))
        bad = rng.random(n) < 0.2
        z[bad] = rng.uniform(0, 600, (int(bad.sum()), 2))
        p3[k, :n], p2[k, :n] = Pw, z
    d3, d2 = torch.from_numpy(p3).cuda(), torch.from_numpy(p2).cuda()
    cnt = torch.full((sets,), n, dtype=torch.int32, device="cuda")
    seeds = np.arange(sets, dtype=np.uint64)
    for _ in range(3):
        ctx.pnp_ransac(d3, d2, cnt, K4, seeds)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pose, mask, ninl = ctx.pnp_ransac(d3, d2, cnt, K4, seeds)
    e1.record()
    torch.cuda.synchronize()
    print("flvis_hip_pnp_ransac, %d sets of %d correspondences (20 %% outliers): %.1f us per launch, mean inliers %.0f" %
          (sets, n, e0.elapsed_time(e1) / 20 * 1e3, float(ninl.float().mean())))
