"""Micro-benchmark of the image-domain kernels at the bench shape (S streams of 640x480). Used under rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import flvis_amd

S = int(os.environ.get('S', 64)); H, W = 480, 640
torch.manual_seed(0)
ctx = flvis_amd.Context(0)
# band-limited random textures generated on the GPU (harness only)
base = torch.rand((S, 1, H // 8 + 2, W // 8 + 2), device='cuda')
img = torch.nn.functional.interpolate(base, size=(H, W), mode='bicubic', align_corners=False)
fine = torch.rand((S, 1, H // 2, W // 2), device='cuda')
img = img + 0.5 * torch.nn.functional.interpolate(fine, size=(H, W), mode='bilinear', align_corners=False)
img = ((img - img.amin()) / (img.amax() - img.amin()) * 255).round().clamp(0, 255).to(torch.uint8).squeeze(1).contiguous()
img2 = torch.roll(img, shifts=(2, 3), dims=(1, 2)).contiguous()
fp = [15, 30, 5, 500, 0.001, 5]

def timeit(name, fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('%-28s %8.3f ms/call  (%7.1f us/stream)' % (name, ms, ms * 1000 / S))
    return ms

timeit('equalize_hist', lambda: ctx.equalize_hist(img))
timeit('pyr_down L0->L1', lambda: ctx.pyr_down(img))
xy, cnt = ctx.feature_dem_detect(img, fp)
print('dem detect counts', cnt[:8].tolist())
timeit('gftt(500)', lambda: ctx.gftt(img, 500, 0.001, 5))
timeit('feature_dem_detect', lambda: ctx.feature_dem_detect(img, fp))
ex = xy[:, :512].double().contiguous()
timeit('feature_dem_redetect', lambda: ctx.feature_dem_redetect(img, fp, ex, cnt))
pts = xy[:, :256].contiguous(); c = torch.clamp(cnt, max=256)
timeit('lk_track (+4 pyr builds)', lambda: ctx.lk_track(img, img2, pts, pts, c))
