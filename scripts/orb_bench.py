#!/usr/bin/env python3
"""ORB extraction + matching timing on one GPU: 64 synthetic 640x480 keyframes per call (the streams one GPU carries),
cv::ORB parameters of vo_loopclosing.cpp:242.  Prints one JSON line; run under rocprofv3 --kernel-trace for per-kernel times.

usage: orb_bench.py [n_img=64] [iters=10]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import flvis_amd
import _synth as S

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = flvis_amd.Context(0)
base = [S.corner_img(480, 640, 200 + i) for i in range(8)]
imgs = torch.from_numpy(np.stack([base[i % 8] for i in range(n_img)])).cuda()
for _ in range(2):
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(imgs, cap=2048)
    pairs, npairs = ctx.orb_match(desc, cnt, desc.roll(1, 0).contiguous(), cnt.roll(1, 0).contiguous(), 0.8)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
e[0].record()
for _ in range(iters):
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(imgs, cap=2048)
e[1].record()
d2, c2 = desc.roll(1, 0).contiguous(), cnt.roll(1, 0).contiguous()
torch.cuda.synchronize()
e[2].record()
for _ in range(iters):
    pairs, npairs = ctx.orb_match(desc, cnt, d2, c2, 0.8)
e[3].record()
torch.cuda.synchronize()
# single-threaded CPU restatement on one image, for scale (test infrastructure, not shipped)
import time
import _oracle as O
t0 = time.time()
wk, wd = O.orb_detect_and_compute(base[0])
t_cpu = time.time() - t0
t_det = e[0].elapsed_time(e[1]) / iters
t_mat = e[2].elapsed_time(e[3]) / iters
print(json.dumps({"n_img": n_img, "detect_and_compute_ms_per_batch": t_det, "keyframes_per_s": n_img / t_det * 1e3,
                  "match_ms_per_batch": t_mat, "pairs_per_s": n_img / t_mat * 1e3,
                  "mean_keypoints": float(cnt.float().mean()), "overflow": int(ovf.sum()),
                  "mean_matches": float(npairs.float().mean()), "cpu_restatement_ms_per_image": t_cpu * 1e3}))
