"""GPU parity tests (run with -m gpu on MI355X): HIP image kernels vs the CPU oracle, bit-exact, through the C ABI."""
import numpy as np
import pytest

import _oracle as O
import _synth as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import flvis_amd
    c = flvis_amd.Context(0)
    yield c
    c.close()


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("h,w,n", [(48, 64, 3), (480, 640, 2), (480, 752, 2)])
def test_equalize_hist_parity(ctx, h, w, n):
    rng = np.random.default_rng(h * w)
    imgs = np.stack([(S.value_noise(h, w, 20 + i) * 170 + rng.integers(0, 60, (h, w))).astype(np.uint8)
                     for i in range(n)])
    imgs[0, :, :] = np.clip(imgs[0].astype(int) // 3 + 90, 0, 255)  # narrow histogram
    out = ctx.equalize_hist(_cuda(imgs)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], O.equalize_hist(imgs[i]))


@pytest.mark.parametrize("h,w,c,n", [(480, 640, 3, 2), (480, 640, 4, 2), (37, 52, 3, 3), (5, 4, 4, 1)])
def test_cvt_bgr_to_gray_parity(ctx, h, w, c, n):
    rng = np.random.default_rng(h * w + c)
    imgs = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
    out = ctx.cvt_bgr_to_gray(_cuda(imgs)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], O.cvt_bgr_to_gray(imgs[i]))


def test_equalize_hist_constant(ctx):
    imgs = np.full((2, 32, 64), 9, np.uint8)
    imgs[1] = 250
    out = ctx.equalize_hist(_cuda(imgs)).cpu().numpy()
    assert np.array_equal(out, imgs)


@pytest.mark.parametrize("h,w,n", [(480, 640, 2), (240, 320, 3), (120, 160, 1), (60, 80, 2), (480, 752, 1), (62, 92, 2)])
def test_pyr_down_parity(ctx, h, w, n):
    imgs = np.stack([S.texture_u8(h, w, 30 + i) for i in range(n)])
    out = ctx.pyr_down(_cuda(imgs)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], O.pyr_down(imgs[i]))


def _bordered(img, bx, by):
    return np.pad(img, ((by, by), (bx, bx)), mode="reflect")  # numpy's "reflect" is BORDER_REFLECT_101


@pytest.mark.parametrize("h,w,n,levels,ingest", [(480, 640, 3, 3, True), (480, 640, 2, 3, False), (480, 752, 2, 3, True), (376, 1248, 1, 3, True),
                                                 (200, 320, 2, 2, True), (131, 208, 2, 2, False), (97, 144, 1, 1, True), (376, 1241, 1, 3, False),
                                                 (480, 1024, 1, 3, True), (64, 64, 2, 1, True)])
def test_pyramid_with_border_parity(ctx, h, w, n, levels, ingest):
    """The tracker's pyramid construction (walking kernels where the geometry allows, tile kernels otherwise, k_pyr_border for the rest):
    every level equals the checker's pyrDown chain, and the physical border around it is its BORDER_REFLECT_101 continuation."""
    if w % 4:
        pytest.skip("flvis_debug_pyramid takes packed rows of w % 4 == 0 (the tracker copies other widths into pitched level-0 buffers first)")
    imgs = np.stack([S.texture_u8(h, w, 70 + i) for i in range(n)])
    got = ctx.debug_pyramid(_cuda(imgs), levels, 32, 24, ingest)
    for i in range(n):
        lvl = imgs[i]
        for l in range(levels + 1):
            if l:
                lvl = O.pyr_down(lvl)
            if got[l] is None:
                continue
            g = got[l][i].cpu().numpy()
            if lvl.shape[0] > 24 and lvl.shape[1] > 32:
                assert np.array_equal(g, _bordered(lvl, 32, 24)), (i, l)
            else:
                assert np.array_equal(g[24:-24, 32:-32], lvl), (i, l)


@pytest.mark.parametrize("h,w", [(480, 640), (96, 128), (35, 64), (17, 80), (480, 1024)])
def test_pyramid_without_border_parity(ctx, h, w):
    imgs = np.stack([S.texture_u8(h, w, 90 + i) for i in range(2)])
    got = ctx.debug_pyramid(_cuda(imgs), 2 if h >= 32 else 1, 0, 0, True)
    for i in range(2):
        lvl = imgs[i]
        for l in range(len(got)):
            if l:
                lvl = O.pyr_down(lvl)
            assert np.array_equal(got[l][i].cpu().numpy(), lvl), (i, l)


@pytest.mark.parametrize("h,w,maxc,q,md", [(96, 128, 50, 0.01, 5), (480, 640, 500, 0.001, 5), (480, 640, 1000, 0.001, 5),
                                          (480, 752, 1000, 0.01, 10), (100, 132, 40, 0.05, 3)])
def test_gftt_parity(ctx, h, w, maxc, q, md):
    n = 3
    imgs = np.stack([S.texture_u8(h, w, 40 + i) for i in range(n)])
    imgs[2, : h // 2] = 128  # half flat image: few corners
    xy, cnt = ctx.gftt(_cuda(imgs), maxc, q, md)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    for i in range(n):
        want = O.gftt(imgs[i], maxc, q, md)
        assert cnt[i] == len(want), (i, cnt[i], len(want))
        assert np.array_equal(xy[i, :cnt[i]], want)


def test_gftt_flat_image_has_no_corners(ctx):
    imgs = np.full((1, 64, 64), 100, np.uint8)
    xy, cnt = ctx.gftt(_cuda(imgs), 10, 0.01, 5)
    assert int(cnt[0]) == 0


@pytest.mark.parametrize("h,w,fp", [(480, 640, [15, 30, 5, 500, 0.001, 5]), (480, 752, [30, 20, 5, 1000, 0.01, 10])])
def test_feature_dem_detect_parity(ctx, h, w, fp):
    n = 2
    imgs = np.stack([S.texture_u8(h, w, 50 + i) for i in range(n)])
    xy, cnt = ctx.feature_dem_detect(_cuda(imgs), fp)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    for i in range(n):
        want = O.dem_detect(imgs[i], fp)
        assert cnt[i] == len(want)
        assert np.array_equal(xy[i, :cnt[i]], want)


@pytest.mark.parametrize("h,w,fp", [(480, 640, [15, 30, 5, 500, 0.001, 5]), (480, 752, [30, 20, 5, 1000, 0.01, 10])])
def test_feature_dem_redetect_parity(ctx, h, w, fp):
    import torch
    n = 3
    imgs = np.stack([S.texture_u8(h, w, 60 + i) for i in range(n)])
    cap = 512
    ex = np.zeros((n, cap, 2), np.float64)
    nex = np.zeros(n, np.int32)
    for i in range(n):
        first = O.dem_detect(imgs[i], fp)
        sel = first[:: (i + 2)].astype(np.float64) + np.array([0.37, -0.21]) * (i + 1)
        ex[i, :len(sel)] = sel
        nex[i] = len(sel)
    nex[2] = 0  # no existing points at all
    xy, cnt = ctx.feature_dem_redetect(_cuda(imgs), fp, _cuda(ex), _cuda(nex))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    for i in range(n):
        want = O.dem_redetect(imgs[i], fp, ex[i, :nex[i]])
        assert cnt[i] == len(want), (i, cnt[i], len(want))
        assert np.array_equal(xy[i, :cnt[i]], want)


def _tiled(h, w, seed, period=16):
    """one 16 x 16 texture patch repeated over the image: every corner recurs with the same 3 x 3 neighbourhood, so FeatureDEM's
    integer-built Harris scores tie by the dozen inside a region"""
    t = S.texture_u8(3 * period, 3 * period, seed)[period:2 * period, period:2 * period]
    return np.tile(t, (h // period + 1, w // period + 1))[:h, :w].copy()


def test_feature_dem_orders_tied_scores_as_std_sort_does(ctx):
    """feature_dem.cpp:170,230 sort the candidates of a region with std::sort, which is not stable: where scores tie, the order is
    what libstdc++'s introsort leaves, and the greedy spacing walk depends on it.  The oracle calls the real std::sort; the device
    restates the algorithm (csrc/dem_sort.hpp, checked against std::sort on the CPU by tests/test_dem_sort.py).  Tiled images make
    regions of 30 .. 80 candidates in a handful of score classes -- a stable order gives another feature set on them."""
    import torch
    fp = [15, 30, 5, 500, 0.001, 5]
    imgs = np.stack([_tiled(480, 640, 7), _tiled(480, 640, 8, 24), _tiled(480, 640, 9, 12)])
    xy, cnt = ctx.feature_dem_detect(_cuda(imgs), fp)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    first = []
    for i in range(len(imgs)):
        want = O.dem_detect(imgs[i], fp)
        first.append(want)
        assert cnt[i] == len(want) and len(want) > 40, (i, cnt[i], len(want))
        assert np.array_equal(xy[i, :cnt[i]], want), i
    cap = 512
    ex = np.zeros((len(imgs), cap, 2), np.float64)
    nex = np.zeros(len(imgs), np.int32)
    for i in range(len(imgs)):
        sel = first[i][::3].astype(np.float64) + np.array([0.37, -0.21])
        ex[i, :len(sel)] = sel
        nex[i] = len(sel)
    xy, cnt = ctx.feature_dem_redetect(_cuda(imgs), fp, _cuda(ex), _cuda(nex))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    for i in range(len(imgs)):
        want = O.dem_redetect(imgs[i], fp, ex[i, :nex[i]])
        assert cnt[i] == len(want) and len(want) > 10, (i, cnt[i], len(want))
        assert np.array_equal(xy[i, :cnt[i]], want), i


def _lk_case(h, w, n, seed, shifts, npts, max_level=10):
    prev = np.zeros((n, h, w), np.uint8)
    nxt = np.zeros((n, h, w), np.uint8)
    nmax = npts + 7
    pp = np.zeros((n, nmax, 2), np.float32)
    init = np.zeros((n, nmax, 2), np.float32)
    cnt = np.zeros(n, np.int32)
    rng = np.random.default_rng(seed)
    for i in range(n):
        dx, dy = shifts[i % len(shifts)]
        prev[i], nxt[i] = S.shifted_pair(h, w, seed + i, dx, dy)
        pts = O.gftt(prev[i], npts, 0.01, 6)
        extra = np.array([[1.5, 2.5], [w - 2.25, h - 3.5], [w / 2 + 0.123, 0.75], [-8.0, 20.0], [w + 40.0, 10.0]],
                         np.float32)  # border / outside points
        pts = np.concatenate([pts[: npts - len(extra)], extra])
        k = len(pts) - (i % 3)
        pp[i, :k] = pts[:k]
        init[i, :k] = pts[:k] + np.array([dx, dy], np.float32) * 0.6 + rng.normal(0, 0.7, (k, 2)).astype(np.float32)
        cnt[i] = k
    return prev, nxt, pp, init, cnt


@pytest.mark.parametrize("h,w,n,seed,max_level,use_initial", [
    (480, 640, 2, 100, 10, True), (480, 640, 2, 110, 5, True), (240, 320, 3, 120, 10, False),
    (480, 752, 1, 130, 10, True), (120, 160, 2, 140, 1, True),
    (376, 1241, 1, 150, 5, True), (203, 322, 2, 160, 10, True)])      # rows that are not dword aligned (KITTI's 1241 x 376)
def test_lk_parity_bit_exact(ctx, h, w, n, seed, max_level, use_initial):
    shifts = [(3.3, -2.1), (-7.6, 5.2), (0.3, 0.2)]
    prev, nxt, pp, init, cnt = _lk_case(h, w, n, seed, shifts, 120)
    out, st = ctx.lk_track(_cuda(prev), _cuda(nxt), _cuda(pp), _cuda(init), _cuda(cnt), max_level=max_level,
                           use_initial=use_initial)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    for i in range(n):
        k = cnt[i]
        want, wst = O.lk(prev[i], nxt[i], pp[i, :k], init[i, :k], max_level=max_level, use_initial=use_initial)
        assert np.array_equal(st[i, :k], wst), (i, np.nonzero(st[i, :k] != wst))
        assert np.array_equal(out[i, :k].view(np.uint32), want.view(np.uint32)), (
            i, np.abs(out[i, :k] - want).max())


def test_lk_empty_and_ragged(ctx):
    prev, nxt, pp, init, cnt = _lk_case(120, 160, 3, 200, [(1.0, 0.5)], 30)
    cnt[1] = 0
    out, st = ctx.lk_track(_cuda(prev), _cuda(nxt), _cuda(pp), _cuda(init), _cuda(cnt))
    out = out.cpu().numpy()
    assert np.array_equal(out[1], init[1])          # untouched
    want, wst = O.lk(prev[2], nxt[2], pp[2, :cnt[2]], init[2, :cnt[2]])
    assert np.array_equal(out[2, :cnt[2]], want)


def _want_corner_response(img):
    """max ordered bits and the sorted candidate keys of cv::goodFeaturesToTrack's response pass, from the oracle's min-eigenvalue map"""
    e = O.min_eigen_map(img)
    h, w = e.shape
    b = e.view(np.uint32).astype(np.uint64)
    ordered = np.where(b >> np.uint64(31), b ^ np.uint64(0xFFFFFFFF), b ^ np.uint64(0x80000000))
    pad = np.full((h + 2, w + 2), -np.inf, np.float32)
    pad[1:-1, 1:-1] = e
    nb = np.full((h, w), -np.inf, np.float32)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if (dy, dx) != (1, 1):
                nb = np.maximum(nb, pad[dy:dy + h, dx:dx + w])
    ismax = (e > 0) & ~(nb > e)
    ismax[0, :] = ismax[-1, :] = False
    ismax[:, 0] = ismax[:, -1] = False
    ys, xs = np.nonzero(ismax)
    keys = ~((ordered[ys, xs] << np.uint64(32)) | (ys * w + xs).astype(np.uint64))
    return int(ordered.max()), np.sort(keys)


@pytest.mark.parametrize("variant,rows", [(0, 0), (1, 0), (2, 60), (2, 37), (2, 8), (2, 500)])
@pytest.mark.parametrize("h,w", [(480, 640), (480, 752), (376, 1244), (96, 128), (61, 64), (9, 8), (100, 60)])
def test_corner_response_parity(ctx, h, w, variant, rows):
    """The response pass of goodFeaturesToTrack on its own, every kernel variant (LDS tiles, strip-mined tiles, the wave walk with
    several chunk heights incl. one chunk pair per image and chunks shorter than the pipeline lag): the maximum and the COMPLETE
    set of 3x3 local maxima (value bits and pixel offset of every candidate) are those of the oracle's min-eigenvalue map -- at
    image sizes whose strips / chunks / tiles end inside, at and beyond the borders."""
    n = 3
    imgs = np.stack([S.texture_u8(h, w, 70 + i) for i in range(n)])
    imgs[1, : h // 2] = 77                     # flat half: zero responses, no candidates there
    imgs[2] = np.where(np.add.outer(np.arange(h), np.arange(w)) % 2 == 0, 255, 0).astype(np.uint8)  # checkerboard: plateaus of equal maxima
    mx, keys = ctx.debug_corner_response(_cuda(imgs), variant, rows, key_cap=max(1024, h * w))
    for i in range(n):
        wmax, wkeys = _want_corner_response(imgs[i])
        assert int(mx[i]) == wmax, (i, hex(int(mx[i])), hex(wmax))
        assert len(keys[i]) == len(wkeys) and np.array_equal(keys[i], wkeys), (i, len(keys[i]), len(wkeys))


def test_corner_response_sqrt_is_correctly_rounded(ctx):
    """ew_sqrt_pos (eig_walk.hip) drops the rescaling and the special-value fix-up of the correctly rounded sqrtf; over EVERY float of
    its domain -- 0 and [2^-96, 4) -- it returns the same bits (8.2e8 arguments, compared on the device)."""
    assert ctx.debug_sqrt_check(0, 1) == 0
    lo, hi = (127 - 96) << 23, 0x40800000
    assert ctx.debug_sqrt_check(lo, hi - lo) == 0
    assert ctx.debug_sqrt_check(0x00800000, 1 << 20) > 0     # (outside the domain the two do differ: the check is not vacuous)


@pytest.mark.parametrize("walk", ["120", "0"])
def test_gftt_parity_with_the_other_response_kernels(walk):
    """The corner-response pass inside flvis_hip_gftt / FeatureDEM is the wave walk with 60 rows per chunk (120 until round 6); FLVIS_EIG_WALK=<rows>
    changes the chunk height, FLVIS_EIG_WALK=0 selects the LDS-tile kernel.  The switch is read once per process, so the parity tests
    of this file are re-run in a child process with it set: the same corners, bit for bit."""
    import os
    import subprocess
    import sys
    if os.environ.get("FLVIS_EIG_STRIP_CHILD"):
        pytest.skip("this is the child run")
    env = dict(os.environ, FLVIS_EIG_WALK=walk, FLVIS_EIG_STRIP_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "gftt_parity or gftt_flat or feature_dem"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and " passed" in out and "failed" not in out, out[-3000:]


def test_gftt_parity_with_the_strip_mined_response_kernel():
    """FLVIS_EIG_STRIP=1 selects k_eig_cand_strip (four Sobel pairs / responses per thread, flvis_amd/csrc/eig_strip.hpp) for the
    corner-response pass; the switch is read once per process, so the goodFeaturesToTrack / FeatureDEM parity tests of this file
    are re-run in a child process with it set: the same corners, bit for bit."""
    import os
    import subprocess
    import sys
    if os.environ.get("FLVIS_EIG_STRIP_CHILD"):
        pytest.skip("this is the child run")
    env = dict(os.environ, FLVIS_EIG_WALK="0", FLVIS_EIG_STRIP="1", FLVIS_EIG_STRIP_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "gftt_parity or gftt_flat or feature_dem"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and " passed" in out and "failed" not in out, out[-3000:]
