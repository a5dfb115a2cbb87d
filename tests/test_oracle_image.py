"""CPU tests: pin the image-domain oracle (oracle/ref_image.cpp) against INDEPENDENT numpy restatements and
closed-form properties.  The reference ships no golden vectors for these stages (parity unpinned)."""
import numpy as np
import pytest

import _geom as G
import _oracle as O
import _synth as S


def np_reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def np_equalize_hist(img):
    hist = np.bincount(img.ravel(), minlength=256)
    i0 = int(np.nonzero(hist)[0][0])
    total = img.size
    if hist[i0] == total:
        return np.full_like(img, i0)
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    cum = np.cumsum(hist) - np.cumsum(hist)[i0]
    lut = np.rint((cum.astype(np.float32) * scale)).clip(0, 255).astype(np.uint8)
    lut[:i0 + 1] = 0
    return lut[img]


def np_pyr_down(img):
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)
    dh, dw = (h + 1) // 2, (w + 1) // 2
    ys = np_reflect101(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)
    xs = np_reflect101(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)
    a = img.astype(np.int64)
    rows = (a[ys] * k[None, :, None]).sum(1)           # [dh, w]
    out = (rows[:, xs] * k[None, None, :]).sum(2)      # [dh, dw]
    return ((out + 128) >> 8).astype(np.uint8)


def np_min_eig(img):
    h, w = img.shape
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    sc = np.float32(1.0 / (255.0 * 12.0))
    fx = dx.astype(np.float32) * sc
    fy = dy.astype(np.float32) * sc
    cov = [np.pad(c, 1, mode="reflect") for c in (fx * fx, fx * fy, fy * fy)]
    sums = []
    for c in cov:
        acc = np.zeros((h, w), np.float32)
        for j in range(3):
            for i in range(3):
                acc = acc + c[j:j + h, i:i + w]
        sums.append(acc)
    a, b, c = sums[0] * np.float32(0.5), sums[1], sums[2] * np.float32(0.5)
    return (a + c) - np.sqrt((a - c) * (a - c) + b * b, dtype=np.float32)


@pytest.mark.parametrize("shape,seed", [((48, 64), 1), ((120, 160), 2), ((61, 95), 3)])
def test_equalize_hist_matches_numpy(shape, seed):
    rng = np.random.default_rng(seed)
    img = (S.value_noise(shape[0], shape[1], seed) * 180 + rng.integers(0, 40, shape)).astype(np.uint8)
    assert np.array_equal(O.equalize_hist(img), np_equalize_hist(img))


def test_equalize_hist_constant_and_two_level():
    img = np.full((32, 32), 77, np.uint8)
    assert np.array_equal(O.equalize_hist(img), img)
    img[:, 16:] = 200
    out = O.equalize_hist(img)
    assert set(np.unique(out)) == {0, 255}


@pytest.mark.parametrize("shape", [(48, 64), (60, 80), (61, 95), (480, 640)])
def test_pyr_down_matches_numpy(shape):
    img = S.texture_u8(shape[0], shape[1], 5)
    assert np.array_equal(O.pyr_down(img), np_pyr_down(img))


def test_pyr_down_constant_is_constant():
    img = np.full((30, 50), 131, np.uint8)
    assert np.all(O.pyr_down(img) == 131)


def test_lk_levels_clamp():
    L = O.lib().ref_lk_num_levels
    assert L(640, 480, 31, 10) == 3   # maxLevel 10 clamps to 3 at 640x480
    assert L(752, 480, 31, 10) == 3
    assert L(640, 480, 31, 5) == 3
    assert L(640, 480, 31, 2) == 2
    assert L(64, 64, 31, 10) == 1 and L(62, 62, 31, 10) == 0


@pytest.mark.parametrize("shape", [(48, 64), (97, 131)])
def test_min_eig_map_matches_numpy(shape):
    img = S.texture_u8(shape[0], shape[1], 7)
    assert np.array_equal(O.min_eigen_map(img), np_min_eig(img))


def np_gftt(img, max_corners, q, min_dist):
    eig = np_min_eig(img)
    h, w = img.shape
    thr = np.float32(np.float64(eig.max()) * q)
    te = np.where(eig > thr, eig, np.float32(0))
    cand = []
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            v = te[y, x]
            if v != 0 and v == te[y - 1:y + 2, x - 1:x + 2].max():
                cand.append((v, y * w + x))
    cand.sort(key=lambda t: (-t[0], -t[1]))
    out = []
    for v, off in cand:
        y, x = divmod(off, w)
        if all((x - a) ** 2 + (y - b) ** 2 >= min_dist * min_dist for a, b in out):
            out.append((x, y))
            if len(out) == max_corners:
                break
    return np.array(out, np.float32).reshape(-1, 2)


@pytest.mark.parametrize("seed,maxc,q,md", [(1, 50, 0.01, 5), (2, 200, 0.001, 5), (3, 30, 0.05, 10)])
def test_gftt_matches_bruteforce(seed, maxc, q, md):
    img = S.texture_u8(96, 128, seed)
    got = O.gftt(img, maxc, q, md)
    want = np_gftt(img, maxc, q, md)
    assert np.array_equal(got, want)


def test_lk_recovers_subpixel_translation():
    for seed, (dx, dy) in enumerate([(3.3, -2.1), (-6.75, 4.5), (0.4, 0.25)]):
        i0, i1 = S.shifted_pair(240, 320, 10 + seed, dx, dy)
        pts = O.gftt(i0, 150, 0.01, 8)
        keep = (pts[:, 0] > 40) & (pts[:, 0] < 280) & (pts[:, 1] > 40) & (pts[:, 1] < 200)
        pts = pts[keep]
        nxt, st = O.lk(i0, i1, pts, pts)
        assert st.mean() > 0.95
        err = np.abs(nxt[st == 1] - pts[st == 1] - np.array([dx, dy], np.float32))
        assert np.median(err) < 0.03
        assert np.percentile(err, 90) < 0.15


def test_lk_status_rules():
    i0, i1 = S.shifted_pair(120, 160, 3, 1.0, 0.5)
    flat = np.full_like(i0, 128)
    pts = np.array([[80, 60], [5, 5], [500, 60]], np.float32)
    _, st = O.lk(flat, flat, pts, pts)              # no texture -> min-eig test fails
    assert st.tolist() == [0, 0, 0]
    _, st = O.lk(i0, i1, pts, pts)
    assert st[0] == 1 and st[2] == 0               # far outside -> window leaves the image at level 0


def test_lk_identity_is_fixed_point():
    i0 = S.texture_u8(120, 160, 4)
    pts = O.gftt(i0, 40, 0.01, 10)
    nxt, st = O.lk(i0, i0, pts, pts)
    assert np.all(st == 1)
    assert np.abs(nxt - pts).max() < 1e-3


def test_feature_dem_detect_invariants():
    img = S.texture_u8(480, 640, 9)
    fp = [15, 30, 5, 500, 0.001, 5]
    pts = O.dem_detect(img, fp)
    assert 100 < len(pts) <= 240
    reg = (4 * (pts[:, 1] // 120) + pts[:, 0] // 160).astype(int)
    assert np.all(np.diff(reg) >= 0)                               # grouped by region 0..15
    for r in range(16):
        p = pts[reg == r]
        assert len(p) <= 15
        for i in range(len(p)):                                    # cross-shaped spacing (quirk A7): both |dx|,|dy| > 2
            d = np.abs(p - p[i])
            d[i] = 99
            assert np.all((d[:, 0] > 2) & (d[:, 1] > 2))
    assert pts[:, 0].min() >= 3 and pts[:, 0].max() < 637 and pts[:, 1].min() >= 3 and pts[:, 1].max() < 477


def test_feature_dem_redetect_respects_existing():
    img = S.texture_u8(480, 640, 9)
    fp = [15, 30, 5, 500, 0.001, 5]
    first = O.dem_detect(img, fp)
    existed = first[::2].astype(np.float64) + 0.37
    new = O.dem_redetect(img, fp, existed)
    assert len(new) > 0
    assert np.all(new == np.round(new))                            # integer cv::Point quirk (A8)
    for p in new:
        same = (existed[:, 1] // 120 == p[1] // 120) & (existed[:, 0] // 160 == p[0] // 160)
        d = np.abs(existed[same].astype(np.float32) - p)
        assert np.all((d[:, 0] > 2) & (d[:, 1] > 2))


def test_cvt_bgr_to_gray_matches_the_fixed_point_formula():
    """cv::cvtColor BGR2GRAY / BGRA2GRAY (f2f_tracking.cpp:74-111): (1868 B + 9617 G + 4899 R + 2^13) >> 14."""
    rng = np.random.default_rng(1)
    for c in (3, 4):
        img = rng.integers(0, 256, (37, 52, c), dtype=np.uint8)
        want = ((img[..., 0].astype(np.int64) * 1868 + img[..., 1].astype(np.int64) * 9617 + img[..., 2].astype(np.int64) * 4899
                 + (1 << 13)) >> 14).astype(np.uint8)
        assert np.array_equal(O.cvt_bgr_to_gray(img), want)
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)       # B = G = R -> unchanged
    assert np.array_equal(O.cvt_bgr_to_gray(grey)[0], np.arange(256))


# ------------------------------------------------------------------------------------------------ FeatureDEM, restated again
def np_harris_r(img, ptx, pty):
    """FeatureDEM::calHarrisR (feature_dem.cpp:59-88) with its quirks: patch[5] is the (x+1, y+1) pixel, integer division
    by 3, Y2 = IY*IX and XY = IX*IX."""
    xx, yy = int(ptx), int(pty)
    p = [int(img[yy - 1, xx - 1]), int(img[yy - 1, xx]), int(img[yy - 1, xx + 1]), int(img[yy, xx - 1]), int(img[yy, xx]),
         int(img[yy + 1, xx + 1]), int(img[yy + 1, xx - 1]), int(img[yy + 1, xx]), int(img[yy + 1, xx + 1])]
    tdiv = lambda a: int(a / 3)                      # C integer division truncates toward zero
    IX = np.float32(tdiv(p[0] + p[3] + p[6] - (p[2] + p[5] + p[8])))
    IY = np.float32(tdiv(p[0] + p[1] + p[2] - (p[6] + p[7] + p[8])))
    X2, Y2, XY = IX * IX, IY * IX, IX * IX
    return np.float32(np.float32(X2 * Y2) - np.float32(XY * XY)) - np.float32(np.float32(np.float32(0.05) * np.float32(X2 + Y2)) * np.float32(X2 + Y2))


def np_fill_regions(img, pts, w, h, existed):
    rw, rh = int(np.floor(w / 4.0)), int(np.floor(h / 4.0))
    regions = [[] for _ in range(16)]
    for (x, y) in pts:
        x, y = np.float32(x), np.float32(y)
        if x >= 3 and x < w - 3 and y >= 3 and y < h - 3:
            r = int(np.float32(4 * np.floor(np.float32(y / np.float32(rh)))) + np.float32(x / np.float32(rw)))
            regions[r].append(((x, y), np.float32(99999.0) if existed else np_harris_r(img, x, y)))
    return regions


def np_dem_detect(img, f_para):
    h, w = img.shape
    bd, maxn = int(np.floor(f_para[2] / 2.0)), int(f_para[0])
    corners = O.gftt(img, 2 * int(f_para[3]), f_para[4], int(f_para[5]))
    out = []
    for reg in np_fill_regions(img, corners, w, h, False):
        reg = G.gnu_sort(list(reg), lambda a, b: a[1] > b[1])  # std::sort(..., sortbysecdesc): ties where libstdc++'s introsort leaves them
        kept = []
        for (pt, _) in reg:
            ok = all(not (abs(pt[0] - k[0]) <= bd or abs(pt[1] - k[1]) <= bd) for k in kept)
            if ok:
                kept.append(pt)
                if len(kept) >= maxn:
                    break
        out += kept
    return np.array(out, np.float32).reshape(-1, 2)


def np_dem_redetect(img, f_para, existed):
    h, w = img.shape
    bd, maxn = int(np.floor(f_para[2] / 2.0)), int(f_para[0])
    regions = np_fill_regions(img, [(np.float32(x), np.float32(y)) for x, y in existed], w, h, True)
    regions = [[pt for pt, _ in reg] for reg in regions]
    corners = O.gftt(img, int(f_para[3]), f_para[4], int(f_para[5]))
    new = []
    for i, reg in enumerate(np_fill_regions(img, corners, w, h, False)):
        for (pt, _) in G.gnu_sort(list(reg), lambda a, b: a[1] > b[1]):
            ip = (int(np.rint(pt[0])), int(np.rint(pt[1])))     # cv::Point pt = Point2f: rounds half to even
            near = any(abs(np.float32(ip[0]) - k[0]) <= bd or abs(np.float32(ip[1]) - k[1]) <= bd for k in regions[i])
            if not near:
                regions[i].append((np.float32(ip[0]), np.float32(ip[1])))
                new.append(ip)
                if len(regions[i]) >= maxn:
                    break
    return np.array(new, np.float32).reshape(-1, 2)


@pytest.mark.parametrize("seed,fp", [(3, [15, 30, 5, 500, 0.001, 5]), (4, [30, 20, 5, 1000, 0.01, 10]), (5, [4, 30, 9, 300, 0.005, 7])])
def test_feature_dem_matches_an_independent_restatement(seed, fp):
    img = S.texture_u8(480, 640, seed)
    got = O.dem_detect(img, fp)
    want = np_dem_detect(img, fp)
    assert len(want) > 30 and got.shape == want.shape and np.array_equal(got, want)
    rng = np.random.default_rng(seed)
    existed = got[rng.permutation(len(got))[: len(got) // 2]].astype(np.float64) + rng.uniform(-0.4, 0.4, (len(got) // 2, 2))
    got2 = O.dem_redetect(img, fp, existed)
    want2 = np_dem_redetect(img, fp, existed)
    assert len(want2) > 5 and got2.shape == want2.shape and np.array_equal(got2, want2)
