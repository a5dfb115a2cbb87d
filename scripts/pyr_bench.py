"""Pyramid construction alone (flvis_debug_pyramid) on a batch of images: run under `rocprofv3 --kernel-trace --stats` for the kernels' durations.
usage: pyr_bench.py [n_img] [reps] [ingest]"""
import sys
import numpy as np
import torch
import flvis_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ingest = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
ctx = flvis_amd.Context()
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.integers(0, 256, (n, 480, 640), dtype=np.uint8)).cuda()
for _ in range(reps):
    ctx.debug_pyramid(img, 3, 32, 24, ingest)
torch.cuda.synchronize()
print("done", n, reps, ingest)
