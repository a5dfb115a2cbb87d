"""The oracle's loop-closing keyframe landmarks (oracle/ref_loopkf.cpp; vo_loopclosing.cpp:255-372) against an independent
composition: the stereo branch from the oracle's own LK + numpy SVD triangulation, the depth branch from plain numpy."""
import numpy as np

import _oracle as O
import _synth as S


def _kps(rng, n, w, h, margin=40):
    k = np.zeros((n, 6), np.float32)
    k[:, 0] = rng.uniform(margin, w - margin, n).astype(np.float32)
    k[:, 1] = rng.uniform(margin, h - margin, n).astype(np.float32)
    return k


def test_stereo_branch_composition():
    h, w = 240, 320
    rng = np.random.default_rng(0)
    i0, i1 = S.shifted_pair(h, w, 3, -6.5, 0.0)           # the second camera sees the scene 6.5 px to the left: one depth
    fx, b = 200.0, 0.05
    P0 = np.array([fx, 0, 160, 0, 0, fx, 120, 0, 0, 0, 1, 0.0])
    P1 = np.array([fx, 0, 160, -fx * b, 0, fx, 120, 0, 0, 0, 1, 0.0])
    kps = _kps(rng, 150, w, h)
    kps[::7, 0] = 3.0                                       # lost by LK at the border / flat -> dropped
    desc = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    lm2, lm3, lmd = O.lc_keyframe_landmarks(i0, i1, 0, kps, desc, P0, P1)
    nxt, st = O.lk(i0, i1, kps[:, :2], kps[:, :2], max_level=5)
    keep = []
    for i in range(len(kps)):
        if st[i] != 1:
            continue
        u1, v1 = [float(x) for x in kps[i, :2]]
        u2, v2 = [float(x) for x in nxt[i]]
        Pa, Pb = P0.reshape(3, 4), P1.reshape(3, 4)
        A = np.stack([v1 * Pa[2] - Pa[1], Pa[0] - u1 * Pa[2], v2 * Pb[2] - Pb[1], Pb[0] - u2 * Pb[2]])
        X = np.linalg.svd(A)[2][-1]
        X = X[:3] / X[3]
        if X[2] < 0 or X[2] > 100.0:
            continue
        keep.append((i, X))
    assert 60 < len(keep) < len(kps)
    idx = np.array([k[0] for k in keep])
    assert len(lm2) == len(keep)
    assert np.array_equal(lm2, kps[idx, :2]) and np.array_equal(lmd, desc[idx])
    want = np.stack([k[1] for k in keep])
    assert np.abs(lm3 - want).max() < 1e-8 * max(1.0, np.abs(want).max())
    z = fx * b / 6.5
    assert np.median(np.abs(lm3[:, 2] - z)) < 0.02 * z


def test_depth_branch_integer_metres():
    h, w = 120, 160
    rng = np.random.default_rng(1)
    depth = rng.integers(0, 14000, (h, w)).astype(np.uint16)
    kps = _kps(rng, 300, w, h, margin=2)
    kps[0, :2] = (10.5, 20.5)       # ties round to even: pixel (10, 20)
    kps[1, :2] = (11.5, 21.5)       # pixel (12, 22)
    depth[20, 10], depth[22, 12] = 2999, 3000
    desc = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    K4 = np.array([150.0, 151.0, 80.0, 60.0])
    lm2, lm3, lmd = O.lc_keyframe_landmarks(None, depth, 2, kps, desc, K4=K4)
    keep, pts = [], []
    for i, (x, y) in enumerate(kps[:, :2]):
        d = float(int(depth[int(np.rint(y)), int(np.rint(x))]) // 1000)       # whole metres: the reference divides two integers
        if 0.3 <= d <= 10:
            keep.append(i)
            pts.append([(float(x) - K4[2]) / K4[0] * d, (float(y) - K4[3]) / K4[1] * d, d])
    assert keep[:2] == [0, 1] and pts[0][2] == 2.0 and pts[1][2] == 3.0
    assert np.array_equal(lm2, kps[keep, :2]) and np.array_equal(lmd, desc[keep]) and np.array_equal(lm3, np.array(pts))
    assert 0 < len(keep) < 300


def test_unrectified_stereo_keeps_nothing_and_empty_input():
    rng = np.random.default_rng(2)
    i0 = S.texture_u8(64, 64, 1)
    kps = _kps(rng, 10, 64, 64, margin=20)
    desc = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    assert len(O.lc_keyframe_landmarks(i0, i0, 1, kps, desc)[0]) == 0
    P = np.array([100.0, 0, 32, 0, 0, 100, 32, 0, 0, 0, 1, 0])
    assert len(O.lc_keyframe_landmarks(i0, i0, 0, kps[:0], desc[:0], P, P)[0]) == 0
