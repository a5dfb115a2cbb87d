#!/usr/bin/env python3
"""LK microbenchmark: k_lk_track latency vs batch size (run under rocprofv3 --kernel-trace to read the kernel times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flvis_amd

dev = torch.device("cuda", 0)
ctx = flvis_amd.Context(0)
torch.manual_seed(1)
h, w, npts = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 300
base = torch.nn.functional.avg_pool2d(torch.rand(1, 1, h + 16, w + 16, device=dev) * 255, 5, 1, 2)[0, 0]
MI = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ML = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for S in (1, 64):
    i0 = base[8:8 + h, 8:8 + w].to(torch.uint8)[None].repeat(S, 1, 1).contiguous()
    i1 = base[7:7 + h, 6:6 + w].to(torch.uint8)[None].repeat(S, 1, 1).contiguous()  # shift (+2, +1)
    g = torch.stack(torch.meshgrid(torch.linspace(40, w - 40, 20, device=dev), torch.linspace(40, h - 40, npts // 20, device=dev),
                                   indexing="xy"), -1).reshape(-1, 2).float()
    pts = g[None].repeat(S, 1, 1).contiguous()
    cnt = torch.full((S,), pts.shape[1], dtype=torch.int32, device=dev)
    for rep in range(3):
        out, st = ctx.lk_track(i0, i1, pts, pts.clone(), cnt, max_level=ML, max_iter=MI)
    torch.cuda.synchronize()
    if hasattr(ctx._lib, "flvis_debug_lk_prof"):
        import ctypes as C
        buf = (C.c_ulonglong * 16)()
        ctx._lib.flvis_debug_lk_prof(buf, 1)
        v = list(buf); npt = max(v[15], 1)
        names = ["tpl_prefetch", "level_setup+region_issue", "tpl:bilinear", "tpl:lds_load(+A_sums)", "tpl:scharr(+minEig+region_store)", "iter_interp", "iter_reduce", "iter_update"]
        print("  points", v[15], "iterations/point %.1f" % (v[14] / npt), " ".join("%s=%.2fus" % (n, v[i] / npt / 2392.0) for i, n in enumerate(names)), "total=%.1fus" % (sum(v[:8]) / npt / 2392.0))
    print("S", S, "tracked", int(st.sum().item()), "of", st.numel(), "mean flow", (out - pts)[st.bool()].mean(0).tolist())
