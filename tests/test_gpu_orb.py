"""GPU parity tests (-m gpu, MI355X) of the ORB extraction + Hamming matching kernels against the CPU
restatement, through the C ABI.  Everything here is integer/byte work or op-by-op defined float arithmetic: bit-exact."""
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


@pytest.mark.parametrize("shape,dsize", [((480, 640), (533, 400)), ((400, 533), (444, 333)), ((161, 214), (179, 134)),
                                         ((480, 752), (627, 400)), ((97, 131), (64, 50)), ((60, 80), (80, 60)),
                                         ((50, 70), (200, 170))])
def test_resize_linear_parity(ctx, shape, dsize):
    imgs = np.stack([S.texture_u8(shape[0], shape[1], 50 + i) for i in range(3)])
    out = ctx.resize_linear(_cuda(imgs), dsize[0], dsize[1]).cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], O.resize_linear(imgs[i], dsize[0], dsize[1]))


@pytest.mark.parametrize("h,w,thr", [(480, 640, 20), (134, 179, 20), (100, 131, 7), (75, 64, 40), (480, 752, 20)])
def test_fast_score_parity(ctx, h, w, thr):
    imgs = np.stack([S.corner_img(h, w, 60 + i) for i in range(2)])
    out = ctx.fast_score(_cuda(imgs), thr).cpu().numpy()
    for i in range(2):
        want = O.fast_score_map(imgs[i], thr)
        assert (want > 0).sum() > 20
        assert np.array_equal(out[i], want)


@pytest.mark.parametrize("h,w", [(480, 640), (134, 179), (75, 101), (16, 20)])
def test_gaussian_blur_parity(ctx, h, w):
    imgs = np.stack([S.texture_u8(h, w, 70 + i) for i in range(2)])
    out = ctx.gaussian_blur7(_cuda(imgs)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], O.gaussian_blur7(imgs[i]))


def _orb_compare(ctx, imgs, **kw):
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(_cuda(imgs), cap=4096, **kw)
    kps, desc, cnt, ovf = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy(), ovf.cpu().numpy()
    okw = {}
    if "nfeatures" in kw:
        okw["nfeatures"] = kw["nfeatures"]
    if "nlevels" in kw:
        okw["nlevels"] = kw["nlevels"]
    if "scale_factor" in kw:
        okw["sf"] = kw["scale_factor"]
    if "fast_threshold" in kw:
        okw["fast_thr"] = kw["fast_threshold"]
    if "pattern" in kw:
        okw["pattern"] = kw["pattern"]
    total = 0
    for i in range(len(imgs)):
        wk, wd = O.orb_detect_and_compute(imgs[i], **okw)
        assert ovf[i] == 0
        assert cnt[i] == len(wk), (i, cnt[i], len(wk))
        g = kps[i, :cnt[i]]
        assert np.array_equal(g[:, [0, 1, 2, 5]], wk[:, [0, 1, 2, 5]]), "keypoint positions / octaves differ"
        assert np.array_equal(g[:, 4].view(np.uint32), wk[:, 4].view(np.uint32)), "Harris responses differ"
        assert np.array_equal(g[:, 3].view(np.uint32), wk[:, 3].view(np.uint32)), "angles differ"
        assert np.array_equal(desc[i, :cnt[i]], wd), "descriptors differ"
        total += len(wk)
    return total


def test_orb_detect_and_compute_parity_reference_parameters(ctx):
    """cv::ORB::create(1000,1.2f,8,31,0,2,HARRIS_SCORE,31,20) of vo_loopclosing.cpp:242 on 640x480."""
    imgs = np.stack([S.corner_img(480, 640, 80), S.corner_img(480, 640, 81), S.texture_u8(480, 640, 82)])
    imgs[2, :240] = 90          # half flat: some levels come up short of their budget
    assert _orb_compare(ctx, imgs) > 1500


def test_orb_detect_and_compute_parity_euroc_size_and_edge_cases(ctx):
    imgs = np.stack([S.corner_img(480, 752, 83), np.full((480, 752), 128, np.uint8)])   # second image: no corner at all
    assert _orb_compare(ctx, imgs) > 500
    # other parameters, a caller-supplied pattern
    rng = np.random.default_rng(4)
    pat = rng.integers(-13, 14, (512, 2)).astype(np.int8)
    small = np.stack([S.corner_img(200, 260, 84)])
    assert _orb_compare(ctx, small, nfeatures=300, nlevels=4, scale_factor=1.5, fast_threshold=12, pattern=pat) > 100


def test_orb_batch_is_consistent_at_full_size(ctx):
    """64 images (BASELINE streams per GPU): identical images give identical results wherever they sit in the batch."""
    base = [S.corner_img(480, 640, 90 + i) for i in range(4)]
    imgs = np.stack([base[i % 4] for i in range(64)])
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(_cuda(imgs), cap=2048)
    kps, desc, cnt = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy()
    assert int(ovf.sum()) == 0 and cnt.min() > 500
    for i in range(4, 64):
        assert cnt[i] == cnt[i % 4]
        assert np.array_equal(kps[i, :cnt[i]], kps[i % 4, :cnt[i]]) and np.array_equal(desc[i, :cnt[i]], desc[i % 4, :cnt[i]])
    wk, wd = O.orb_detect_and_compute(base[3])
    assert cnt[63] == len(wk) and np.array_equal(desc[63, :cnt[63]], wd)


def _desc_sets(rng, na, nb, nshared):
    a = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    k = min(nshared, na, nb)
    if k:
        b[:k] = a[na - k:]
        noise = (rng.integers(0, 256, (k, 32)) < 14) * (1 << rng.integers(0, 8, (k, 32)))
        b[:k] ^= noise.astype(np.uint8)
    if nb > k + 1 and k > 2:
        b[k] = b[1]                     # exact duplicate: ties keep the lower index
    return a, b


def test_hamming_knn2_and_match_parity(ctx):
    import torch
    rng = np.random.default_rng(7)
    sizes = [(1000, 1000, 600), (700, 1024, 300), (1, 5, 0), (0, 10, 0), (300, 1, 0), (257, 513, 200), (40, 2, 2)]
    acap, bcap = 1000, 1024
    A = np.zeros((len(sizes), acap, 32), np.uint8)
    B = np.zeros((len(sizes), bcap, 32), np.uint8)
    na = np.array([s[0] for s in sizes], np.int32)
    nb = np.array([s[1] for s in sizes], np.int32)
    sets = []
    for p, (n1, n2, k) in enumerate(sizes):
        a, b = _desc_sets(rng, n1, n2, k)
        A[p, :n1], B[p, :n2] = a, b
        sets.append((a, b))
    idx, dist = ctx.hamming_knn2(_cuda(A), _cuda(na), _cuda(B), _cuda(nb))
    pairs, npairs = ctx.orb_match(_cuda(A), _cuda(na), _cuda(B), _cuda(nb), 0.8)
    torch.cuda.synchronize()
    idx, dist, pairs, npairs = idx.cpu().numpy(), dist.cpu().numpy(), pairs.cpu().numpy(), npairs.cpu().numpy()
    for p, (a, b) in enumerate(sets):
        if len(a) and len(b):
            wi, wd = O.hamming_knn2(a, b)
            assert np.array_equal(idx[p, :len(a)], wi) and np.array_equal(dist[p, :len(a)], wd)
        want = O.orb_match(a, b, 0.8) if len(a) else np.zeros((0, 2), np.int32)
        assert npairs[p] == len(want), (p, npairs[p], len(want))
        assert np.array_equal(pairs[p, :npairs[p]], want)
    assert npairs[0] > 400


def test_orb_self_match_is_identity(ctx):
    """an image matched against itself: every keypoint with a unique descriptor pairs with itself."""
    import torch
    img = np.stack([S.corner_img(480, 640, 95)])
    kps, desc, cnt, _ = ctx.orb_detect_and_compute(_cuda(img), cap=2048)
    pairs, npairs = ctx.orb_match(desc, cnt, desc, cnt, 0.8)
    torch.cuda.synchronize()
    n = int(npairs[0])
    pr = pairs[0, :n].cpu().numpy()
    assert n > 0.8 * int(cnt[0]) and np.array_equal(pr[:, 0], pr[:, 1]) and np.all(np.diff(pr[:, 0]) > 0)
