"""CPU tests of the ORB / Hamming-matching restatement (oracle/ref_orb.cpp) against independent numpy
formulations of the published definitions (OpenCV itself is not available: parity unpinned, see the oracle's header)."""
import numpy as np
import pytest
import scipy.ndimage as ndi

import _oracle as O
import _synth as S

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
          (-3, 1), (-2, 2), (-1, 3)]


corner_img = S.corner_img


def np_fast_score(img, thr):
    """score = (largest t with 9 contiguous circle pixels all > t darker, or all > t brighter) - 1, 0 if that t <= thr."""
    h, w = img.shape
    im = img.astype(np.int32)
    d = np.zeros((16, h - 6, w - 6), np.int32)
    for k, (dx, dy) in enumerate(CIRCLE):
        d[k] = im[3:h - 3, 3:w - 3] - im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
    best = np.full((h - 6, w - 6), -10 ** 6, np.int32)
    for s in range(16):
        arc = np.stack([d[(s + j) % 16] for j in range(9)])
        best = np.maximum(best, arc.min(0))        # darker arc: min d
        best = np.maximum(best, (-arc).min(0))     # brighter arc: min -d
    out = np.zeros((h, w), np.int32)
    out[3:h - 3, 3:w - 3] = np.where(best > thr, best - 1, 0)
    return out.astype(np.uint8)


@pytest.mark.parametrize("seed,thr", [(1, 20), (2, 7), (3, 40)])
def test_fast_score_map_matches_definition(seed, thr):
    img = corner_img(120, 160, seed)
    got = O.fast_score_map(img, thr)
    want = np_fast_score(img, thr)
    assert (got > 0).sum() > 50
    assert np.array_equal(got, want)


def test_fast_detect_is_strict_3x3_maximum_in_raster_order():
    img = corner_img(150, 200, 5)
    sc = O.fast_score_map(img, 20).astype(np.int32)
    kp = O.fast_detect(img, 20)
    assert len(kp) > 30
    keys = kp[:, 1] * 10000 + kp[:, 0]
    assert np.all(np.diff(keys) > 0)
    mx = ndi.maximum_filter(sc, size=3, mode="constant")
    fp = np.ones((3, 3), bool)
    fp[1, 1] = False
    nb = ndi.maximum_filter(sc, footprint=fp, mode="constant")
    want = np.argwhere((sc > 0) & (sc > nb))
    assert np.array_equal(want[:, ::-1], kp[:, :2]) and np.array_equal(sc[kp[:, 1], kp[:, 0]], kp[:, 2])
    assert mx.max() == sc.max()


@pytest.mark.parametrize("shape,dsize", [((480, 640), (533, 400)), ((400, 533), (444, 333)), ((97, 131), (64, 50)),
                                         ((60, 80), (80, 60))])
def test_resize_linear_close_to_float_bilinear(shape, dsize):
    img = S.texture_u8(shape[0], shape[1], 3)
    dw, dh = dsize
    got = O.resize_linear(img, dw, dh).astype(np.float64)
    sx, sy = shape[1] / dw, shape[0] / dh
    fx = np.clip((np.arange(dw) + 0.5) * sx - 0.5, 0, shape[1] - 1)
    fy = np.clip((np.arange(dh) + 0.5) * sy - 0.5, 0, shape[0] - 1)
    want = ndi.map_coordinates(img.astype(np.float64), np.meshgrid(fy, fx, indexing="ij"), order=1, mode="nearest")
    assert np.max(np.abs(got - want)) <= 1.0 + 1e-9          # 11-bit coefficients + two truncating shifts
    assert np.array_equal(O.resize_linear(img, shape[1], shape[0]), img)


def test_gaussian_blur_fixed_point():
    k = O.gauss_kernel7_fixed()
    assert list(k) == [18, 34, 49, 55, 49, 34, 18]
    img = S.texture_u8(75, 101, 9)
    t = ndi.correlate1d(img.astype(np.int64), k.astype(np.int64), axis=1, mode="mirror")
    t = ndi.correlate1d(t, k.astype(np.int64), axis=0, mode="mirror")
    want = np.clip((t + (1 << 15)) >> 16, 0, 255).astype(np.uint8)
    assert np.array_equal(O.gaussian_blur7(img), want)
    const = np.full((20, 30), 200, np.uint8)
    assert np.all(O.gaussian_blur7(const) == (200 * 257 * 257 + 32768) >> 16)     # the rounded kernel sums to 257


def test_fast_atan2_accuracy_and_quadrants():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 10 ** rng.uniform(-2, 6)
        a = O.fast_atan2(np.float32(y), np.float32(x))
        w = np.degrees(np.arctan2(np.float32(y), np.float32(x))) % 360.0
        assert min(abs(a - w), 360 - abs(a - w)) < 0.02
    assert O.fast_atan2(0.0, 1.0) == 0.0 and abs(O.fast_atan2(1.0, 0.0) - 90) < 1e-4 and abs(O.fast_atan2(0.0, -1.0) - 180) < 1e-4
    assert abs(O.fast_atan2(-1.0, 0.0) - 270) < 1e-4


def test_umax_is_a_disc_and_ic_angle_matches_moments():
    u = O.orb_umax(15)
    assert list(u) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    img = corner_img(90, 90, 11)
    for (x, y) in [(40, 45), (33, 31), (58, 50)]:
        m10 = m01 = 0
        for v in range(-15, 16):
            for uu in range(-u[abs(v)], u[abs(v)] + 1):
                m10 += uu * int(img[y + v, x + uu])
                m01 += v * int(img[y + v, x + uu])
        want = np.degrees(np.arctan2(m01, m10)) % 360
        got = O.orb_ic_angle(img, x, y)
        assert min(abs(got - want), 360 - abs(got - want)) < 0.02


def test_harris_response_matches_formula():
    img = corner_img(60, 60, 4).astype(np.int64)
    for (x, y) in [(20, 20), (31, 27), (40, 35)]:
        a = b = c = 0
        for j in range(y - 3, y + 4):
            for i in range(x - 3, x + 4):
                ix = (img[j, i + 1] - img[j, i - 1]) * 2 + (img[j - 1, i + 1] - img[j - 1, i - 1]) + (img[j + 1, i + 1] - img[j + 1, i - 1])
                iy = (img[j + 1, i] - img[j - 1, i]) * 2 + (img[j + 1, i - 1] - img[j - 1, i - 1]) + (img[j + 1, i + 1] - img[j - 1, i + 1])
                a += ix * ix
                b += iy * iy
                c += ix * iy
        s = (1.0 / (4 * 7 * 255.0)) ** 4
        want = (float(a) * b - float(c) * c - 0.04 * float(a + b) ** 2) * s
        got = O.orb_harris(img.astype(np.uint8), x, y)
        assert abs(got - want) <= 2e-5 * max(1e-12, abs(float(a) * b * s) + abs(float(c) * c * s))


def test_level_geometry_and_feature_budget():
    lw, lh, ls = O.orb_level_sizes(640, 480)
    assert list(lw) == [640, 533, 444, 370, 309, 257, 214, 179] and list(lh) == [480, 400, 333, 278, 231, 193, 161, 134]
    assert np.allclose(ls, 1.2 ** np.arange(8), rtol=1e-6)
    n = O.orb_features_per_level(1000, 8, 1.2)
    assert n.sum() == 1000 and list(n[:3]) == [217, 181, 151] and np.all(np.diff(n) <= 0)


def test_default_pattern_is_the_documented_generator():
    p = O.orb_default_pattern()
    assert p.shape == (512, 2) and p.min() == -15 and p.max() == 15
    state = 0x34985739
    out = []
    for _ in range(8):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append((state & 0xFFFFFFFF) % 31 - 15)
    assert list(p.reshape(-1)[:8]) == out


def test_detect_and_compute_invariants_and_descriptor_recomputation():
    img = corner_img(480, 640, 21)
    kps, desc, lv, bl = O.orb_detect_and_compute(img, want_pyr=True)
    npl = O.orb_features_per_level()
    lw, lh, ls = O.orb_level_sizes(640, 480)
    assert 600 <= len(kps) <= 1100
    octv = kps[:, 5].astype(int)
    assert np.all(np.diff(octv) >= 0)
    pat = O.orb_default_pattern()
    rng = np.random.default_rng(0)
    for l in range(8):
        sel = np.where(octv == l)[0]
        assert len(sel) >= 1 and len(sel) <= npl[l] + 8           # ties can add a few
        assert np.array_equal(lv[l], img if l == 0 else O.resize_linear(lv[l - 1], lw[l], lh[l]))
        assert np.array_equal(bl[l], O.gaussian_blur7(lv[l]))
        x = np.rint(kps[sel, 0] / ls[l]).astype(int)
        y = np.rint(kps[sel, 1] / ls[l]).astype(int)
        assert x.min() >= 31 and x.max() < lw[l] - 31 and y.min() >= 31 and y.max() < lh[l] - 31
        assert np.all(np.diff(y * 100000 + x) > 0)                 # raster order within the level
        assert np.allclose(kps[sel, 2], 31 * ls[l])
        # every keypoint is a FAST non-max-suppressed corner of its level with the Harris response the oracle reports
        fk = {(a, b) for a, b, _ in O.fast_detect(lv[l], 20)}
        assert all((a, b) in fk for a, b in zip(x, y))
        # all rejected border-valid corners have response <= the weakest kept one, or failed the FAST-score cut
        for i in rng.choice(sel, min(6, len(sel)), replace=False):
            cx, cy = int(round(kps[i, 0] / ls[l])), int(round(kps[i, 1] / ls[l]))
            assert kps[i, 4] == O.orb_harris(lv[l], cx, cy) and kps[i, 3] == O.orb_ic_angle(lv[l], cx, cy)
            ang = np.float32(kps[i, 3]) * np.float32(np.pi / np.float32(180.0))
            a, b = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
            bits = []
            for t in range(256):
                vals = []
                for q in (2 * t, 2 * t + 1):
                    px, py = np.float32(pat[q, 0]), np.float32(pat[q, 1])
                    xx = np.float32(px * a) - np.float32(py * b)
                    yy = np.float32(px * b) + np.float32(py * a)
                    vals.append(int(bl[l][cy + int(np.rint(yy)), cx + int(np.rint(xx))]))
                bits.append(1 if vals[0] < vals[1] else 0)
            want = np.packbits(np.array(bits, np.uint8).reshape(32, 8)[:, ::-1], axis=1).reshape(32)
            assert np.array_equal(desc[i], want)


def test_descriptors_track_a_translated_image():
    """the same scene shifted by whole pixels: level-0 keypoints move with it and keep their descriptors."""
    base = corner_img(300, 400, 8)
    a, b = base[10:250, 20:340], base[14:254, 27:347]           # b = a shifted by (-7, -4)
    ka, da = O.orb_detect_and_compute(a, nfeatures=400)
    kb, db = O.orb_detect_and_compute(b, nfeatures=400)
    pa = {(int(k[0]), int(k[1])): i for i, k in enumerate(ka) if k[5] == 0}
    hits = same = 0
    for j, k in enumerate(kb):
        if k[5] != 0:
            continue
        i = pa.get((int(k[0]) + 7, int(k[1]) + 4))
        if i is not None:
            hits += 1
            same += int(np.array_equal(da[i], db[j]) and ka[i][3] == k[3])
    assert hits >= 40 and same == hits


def np_hamming(a, b):
    return (np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2)).sum(2)


def test_hamming_knn2_and_mutual_ratio_match_numpy():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (170, 32), dtype=np.uint8)
    b[:60] = a[40:100]
    flip = rng.integers(0, 256, (60, 32)) < 12
    b[:60] ^= (flip * (1 << rng.integers(0, 8, (60, 32)))).astype(np.uint8)
    b[60] = b[3]                                                 # an exact duplicate: ties must keep the lower index
    idx, dist = O.hamming_knn2(a, b)
    D = np_hamming(a, b)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order) and np.array_equal(dist, np.take_along_axis(D, order, 1))
    pairs = O.orb_match(a, b, 0.8)
    o21 = np.argsort(D.T, axis=1, kind="stable")[:, 0]
    want = [(i, order[i, 0]) for i in range(len(a))
            if o21[order[i, 0]] == i and D[i, order[i, 1]] > 0 and D[i, order[i, 0]] / D[i, order[i, 1]] < np.float32(0.8)]
    assert len(want) >= 50 and [tuple(p) for p in pairs] == want
    assert len(O.orb_match(a[:1], b, 0.8)) == 0
