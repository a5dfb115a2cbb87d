"""GPU parity of the place-recognition half of the loop closing through the C ABI: DBoW3 bag of words of a keyframe's
descriptors, one row of the L1 similarity matrix -- bit-exact against the CPU oracle (word ids, fp64 values and scores: the sums
run in DBoW3's order on both sides)."""
import numpy as np
import pytest

import _oracle as O
import _voc as V
from test_oracle_bow import RefVoc, ref_score

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import flvis_amd
    c = flvis_amd.Context(0)
    yield c
    c.close()


def _batch(kfs, dcap):
    import torch
    n = len(kfs)
    d = np.zeros((n, dcap, 32), np.uint8)
    cnt = np.zeros(n, np.int32)
    for i, k in enumerate(kfs):
        m = min(len(k), dcap)
        d[i, :m] = k[:m]
        cnt[i] = m
    return torch.from_numpy(d).cuda(), torch.from_numpy(cnt).cuda()


def test_bow_transform_parity_bit_exact(ctx):
    kfs = V.make_keyframes(3, n_img=20)
    voc = V.build_vocabulary(kfs[:12])
    assert (voc[3][voc[4] >= 0] == 0).sum() >= 0
    rv = RefVoc(voc)
    ctx.bow_set_vocabulary(*voc)
    # edge cases: an empty keyframe, one descriptor repeated (a single word), a keyframe that fills the capacity exactly
    kfs = kfs + [np.zeros((0, 32), np.uint8), np.repeat(kfs[0][:1], 9, axis=0), np.concatenate([kfs[1], kfs[2]])[:512]]
    desc, cnt = _batch(kfs, 512)
    ids, vals, nnz = ctx.bow_transform(desc, cnt, vcap=512)
    ids, vals, nnz = ids.cpu().numpy(), vals.cpu().numpy(), nnz.cpu().numpy()
    for i, k in enumerate(kfs):
        wi, wv = rv.transform(k[:512])
        assert nnz[i] == len(wi), i
        assert np.array_equal(ids[i, :nnz[i]], wi), i
        assert np.array_equal(vals[i, :nnz[i]], wv), (i, np.abs(vals[i, :nnz[i]] - wv).max())
    assert nnz[20] == 0 and nnz[21] <= 1 and nnz[:20].min() > 5


def test_bow_score_row_parity_bit_exact(ctx):
    import torch
    kfs = V.make_keyframes(4, n_img=40, per_img=(250, 400))
    voc = V.build_vocabulary(kfs[:20], k=8, depth=3)
    rv = RefVoc(voc)
    ctx.bow_set_vocabulary(*voc)
    desc, cnt = _batch(kfs, 512)
    ids, vals, nnz = ctx.bow_transform(desc, cnt, vcap=512)
    absent = [3, 17]
    db_nnz = nnz.clone()
    db_nnz[absent] = -1                                     # kf_lc_tmp[i] == nullptr
    q = len(kfs) - 1
    scores = ctx.bow_score(ids[q], vals[q], nnz[q:q + 1], ids, vals, db_nnz).cpu().numpy()
    vecs = [rv.transform(k) for k in kfs]
    for j in range(len(kfs)):
        want = 0.0 if j in absent else ref_score(vecs[q], vecs[j])
        assert scores[j] == want, (j, scores[j], want)
    assert abs(scores[q] - 1.0) < 1e-12 and scores[q - 1] > scores[q - 15]


def test_orb_to_bow_chain(ctx):
    """descriptors straight from the ORB kernel (device buffers, no host round trip) through the bag of words"""
    import torch
    from flvis_amd import synth
    tr = [synth.Trajectory(s) for s in range(6)]
    rnd = synth.Renderer("cuda")
    i0, _ = rnd.stereo_frame(tr, 0.5, 10)
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(i0, cap=1024)
    hd, hc = desc.cpu().numpy(), cnt.cpu().numpy()
    assert hc.min() > 200
    train = [hd[i, :hc[i]] for i in range(6)]
    voc = V.build_vocabulary(train, k=8, depth=3)
    rv = RefVoc(voc)
    ctx.bow_set_vocabulary(*voc)
    ids, vals, nnz = ctx.bow_transform(desc, cnt, vcap=1024)
    ids, vals, nnz = ids.cpu().numpy(), vals.cpu().numpy(), nnz.cpu().numpy()
    for i in range(6):
        wi, wv = rv.transform(train[i])
        assert nnz[i] == len(wi) and np.array_equal(ids[i, :nnz[i]], wi) and np.array_equal(vals[i, :nnz[i]], wv), i
