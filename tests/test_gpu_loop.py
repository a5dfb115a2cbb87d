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


def test_bow_vocabulary_from_a_dbow3_file(ctx, tmp_path):
    """`Vocabulary voc(path)` (vo_loopclosing.cpp:1097): the golden file the reference's QuickLZ compressed, loaded by the library,
    gives the vectors the oracle computes on the tree that was written; the other layouts of the same tree give the same."""
    import os
    import flvis_amd
    import _vocfile as VF
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gold, "voc_k6.npz"))
    voc = (z["child_ptr"], z["child_idx"], z["desc"], z["weight"], z["word_id"])
    rv = RefVoc(voc)
    kfs = V.make_keyframes(2024, n_img=12, n_proto=50, per_img=(150, 250))
    desc, cnt = _batch(kfs, 256)
    paths = [os.path.join(gold, "voc_k6_quicklz.dbow3"), str(tmp_path / "v.yml.gz"), str(tmp_path / "v.dbow3")]
    VF.write_yaml(paths[1], voc, 6, 3, gz=True)
    VF.write_binary(paths[2], voc, 6, 3)
    for path in paths:
        ctx.bow_set_vocabulary(*V.build_vocabulary(kfs[:4], k=3, depth=2))     # something else resident first
        ctx.bow_load_vocabulary(path)
        ids, vals, nnz = [t.cpu().numpy() for t in ctx.bow_transform(desc, cnt, vcap=256)]
        for i, k in enumerate(kfs):
            wi, wv = rv.transform(k[:256])
            assert nnz[i] == len(wi) and np.array_equal(ids[i, :nnz[i]], wi) and np.array_equal(vals[i, :nnz[i]], wv), (path, i)
    with pytest.raises(flvis_amd.FlvisError) as e:
        ctx.bow_load_vocabulary(str(tmp_path / "absent.dbow3"))
    assert "cannot open" in str(e.value)
    VF.write_binary(paths[2], voc, 6, 3, scoring=3)
    with pytest.raises(flvis_amd.FlvisError) as e:
        ctx.bow_load_vocabulary(paths[2])
    assert "scoring type 3" in str(e.value)


def test_bow_values_match_the_reference_bowvector_class(ctx):
    """pinned on reference code: golden vectors built by DBoW3's own BowVector class (tests/golden/bowvector_ref.npz)"""
    cases, word_weight = V.bowvector_golden()
    voc, leaf = V.flat_vocabulary(word_weight)
    ctx.bow_set_vocabulary(*voc)
    desc, cnt = _batch([leaf[w] for w, _, _ in cases], 2048)
    ids, vals, nnz = [t.cpu().numpy() for t in ctx.bow_transform(desc, cnt, vcap=512)]
    for i, (words, wi, wv) in enumerate(cases):
        assert nnz[i] == len(wi) and np.array_equal(ids[i, :nnz[i]], wi) and np.array_equal(vals[i, :nnz[i]], wv), (i, len(words))


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
    # several rows of one store in one launch (the loop closer's rows of all sequences; disjoint output ranges): the same numbers
    jobs = [(39, 0, 20), (20, 25, 10), (7, 38, 1), (3, 36, 0)]
    got = ctx.bow_score_jobs(jobs, ids, vals, nnz).cpu().numpy()
    for qv, first, n in jobs:
        assert np.array_equal(got[first:first + n], [ref_score(vecs[qv], vecs[j]) for j in range(first, first + n)]), (qv, first, n)
    assert np.all(got[20:25] == -1) and np.all(got[35:38] == -1) and got[39] == -1


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


def test_pose_graph_optimisation_parity(ctx):
    """loopClosureOnCovGraphG2ONew on the GPU (one workgroup per pose graph, a batch of graphs in one launch) against the CPU
    oracle: the same graph, the same Levenberg decisions; fp64 with other summation orders (block-profile Cholesky by one wave vs
    the scalar profile Cholesky of the restatement): optimised poses within 1e-8."""
    import _pgo_synth as PS
    from test_oracle_pgo import pgo as ref_pgo
    cases = [PS.make_loop(3), PS.make_loop(5, n_kf=90, extra_loops=2), PS.make_loop(7, n_kf=40), PS.make_loop(9, n_kf=130, extra_loops=1)]
    present = [np.ones(len(c["est"]), np.uint8) for c in cases]
    present[1][[20, 21, 47]] = 0
    # a graph without loops, and one whose loop names an absent keyframe: both are left alone
    cases.append(dict(est=cases[0]["est"].copy(), loops=np.zeros((0, 2), np.int32), loop_poses=np.zeros((0, 7))))
    present.append(np.ones(len(cases[0]["est"]), np.uint8))
    cases.append(dict(est=cases[2]["est"].copy(), loops=cases[2]["loops"], loop_poses=cases[2]["loop_poses"]))
    bad = np.ones(len(cases[2]["est"]), np.uint8)
    bad[cases[2]["loops"][0][1]] = 0
    present.append(bad)
    worst = 0.0
    for guess in (True, False):
        got, drift, stats, ran = ctx.pgo_loop_closure([c["est"] for c in cases], present, [c["loops"] for c in cases],
                                                      [c["loop_poses"] for c in cases], use_initial_guess=guess)
        assert list(ran) == [1, 1, 1, 1, 0, 0]
        for k, c in enumerate(cases):
            r, want, wdrift, wstats = ref_pgo(c["est"], present[k], c["loops"], c["loop_poses"], initial_guess=guess)
            assert r == ran[k]
            if not r:
                assert np.array_equal(got[k], c["est"])
                continue
            assert stats[k][3] == wstats[3] and stats[k][4] == wstats[4]
            # (at convergence the chi2 decrease is rounding noise, so the iteration at which Levenberg stops -- rho == 0 or ten failed
            # trials -- is not comparable; the optimum is)
            assert stats[k][0] >= 3 and wstats[0] >= 3, (k, stats[k], wstats)
            assert abs(stats[k][1] - wstats[1]) < 1e-9 * max(1.0, wstats[1]) and abs(stats[k][2] - wstats[2]) < 1e-9
            assert np.abs(got[k] - want).max() < 1e-8, (k, guess, np.abs(got[k] - want).max())
            assert np.abs(drift[k] - wdrift).max() < 1e-8
            assert wstats[2] < 0.2 * wstats[1]
            worst = max(worst, np.abs(got[k] - want).max())
    print("worst pose difference vs the oracle: %.3g" % worst)


def test_pnp_ransac_sets_parity(ctx):
    """flvis_hip_pnp_ransac (the geometric check of isLoopClosureKF: solvePnPRansac P3P, 100 iterations, 2.0 px, 0.99) against the
    CPU oracle's solver on the same correspondences and seeds: the same winning hypothesis, the same inlier mask, the same refined pose
    bit for bit (it is the tracker's solver, whose sums run in the oracle's order)."""
    import torch
    import _geom as G
    rng = np.random.default_rng(8)
    K4 = np.array([384.0, 385.0, 320.0, 240.0])
    cap, sets = 700, []
    for n, outl in ((650, 0.3), (120, 0.1), (40, 0.5), (3, 0.0), (0, 0.0), (700, 0.6)):
        R = G.rodrigues(rng.normal(0, 0.3, 3))
        t = rng.normal(0, 0.5, 3)
        P = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2, 8, n)], 1)
        X = P @ R.T + t
        uv = np.stack([K4[0] * X[:, 0] / X[:, 2] + K4[2], K4[1] * X[:, 1] / X[:, 2] + K4[3]], 1) + rng.normal(0, 0.4, (n, 2))
        bad = rng.random(n) < outl
        uv[bad] = np.stack([rng.uniform(0, 640, bad.sum()), rng.uniform(0, 480, bad.sum())], 1)
        sets.append((P.astype(np.float32), uv.astype(np.float32), R, t, bad))
    p3 = np.zeros((len(sets), cap, 3), np.float32)
    p2 = np.zeros((len(sets), cap, 2), np.float32)
    cnt = np.zeros(len(sets), np.int32)
    for k, (P, uv, _, _, _) in enumerate(sets):
        p3[k, :len(P)], p2[k, :len(P)], cnt[k] = P, uv, len(P)
    seeds = np.array([0x1234 + 77 * k for k in range(len(sets))], np.uint64)
    pose, mask, ninl = ctx.pnp_ransac(torch.from_numpy(p3).cuda(), torch.from_numpy(p2).cuda(), torch.from_numpy(cnt).cuda(), K4, seeds)
    pose, mask, ninl = pose.cpu().numpy(), mask.cpu().numpy(), ninl.cpu().numpy()
    for k, (P, uv, R, t, bad) in enumerate(sets):
        n_want, pose_want, mask_want = O.solve_pnp_ransac(P, uv, K4, iterative=False, iterations=100, reproj=2.0, conf=0.99, seed=int(seeds[k]))
        assert ninl[k] == n_want, (k, ninl[k], n_want)
        assert np.array_equal(mask[k, :len(P)], mask_want) and not mask[k, len(P):].any(), k
        assert np.array_equal(pose[k], pose_want), (k, pose[k] - pose_want)
        if len(P) >= 40:
            Rg, tg = G.pose7_to_Rt(pose[k])
            assert np.degrees(np.arccos(np.clip((np.trace(Rg @ R.T) - 1) / 2, -1, 1))) < 0.3 and np.linalg.norm(tg - t) < 0.02, k
            assert (mask[k, :len(P)][~bad] == 1).mean() > 0.8 and (mask[k, :len(P)][bad] == 1).mean() < 0.05   # (the mask is the winning P3P hypothesis', before the refinement)
    assert ninl[3] == 0 and ninl[4] == 0                      # fewer than four correspondences: no model


def test_loop_verification_chain_on_device(ctx):
    """isLoopClosureKF (vo_loopclosing.cpp:593-700) with the device kernels, no host step in between: ORB of both keyframes, their 3-D
    positions and compacted lists (flvis_hip_lc_keyframe_landmarks), knn x2 + mutual/ratio test on the compacted descriptors,
    solvePnPRansac on (3-D of the earlier keyframe, pixel in the current one), the acceptance rule -- each stage beside the oracle's."""
    import os
    import tempfile
    import torch
    import _geom as G
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_loop_gpu.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
    tr = synth.Trajectory(5)
    rnd = synth.Renderer("cuda")
    ta, tb = 1.0, 1.6
    a0, a1 = [x[0] for x in rnd.stereo_frame([tr], ta, 20)]
    b0, b1 = [x[0] for x in rnd.stereo_frame([tr], tb, 32)]
    i0, i1 = torch.stack([a0, b0]), torch.stack([a1, b1])
    kps, desc, cnt, ovf = ctx.orb_detect_and_compute(i0, cap=1024)
    hk, hd, hc = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy()
    ka, da = O.orb_detect_and_compute(a0.cpu().numpy())
    kb, db = O.orb_detect_and_compute(b0.cpu().numpy())
    assert hc[0] == len(ka) and hc[1] == len(kb) and np.array_equal(hd[0, :hc[0]], da) and np.array_equal(hd[1, :hc[1]], db)
    # KeyFrameLC of both: lm_2d / lm_3d / lm_descriptor without the keypoints that have no 3-D position
    lm2, lm3, lmd, lmc = ctx.lc_keyframe_landmarks(i0, i1, 0, kps, desc, cnt, P0=P0, P1=P1)
    h2, h3, hdd, hcc = lm2.cpu().numpy(), lm3.cpu().numpy(), lmd.cpu().numpy(), lmc.cpu().numpy()
    wa = O.lc_keyframe_landmarks(a0.cpu().numpy(), a1.cpu().numpy(), 0, ka, da, P0, P1)
    wb = O.lc_keyframe_landmarks(b0.cpu().numpy(), b1.cpu().numpy(), 0, kb, db, P0, P1)
    for s, wk in enumerate((wa, wb)):
        assert hcc[s] == len(wk[0]) and np.array_equal(h2[s, :hcc[s]], wk[0]) and np.array_equal(hdd[s, :hcc[s]], wk[2])
        assert np.abs(h3[s, :hcc[s]] - wk[1]).max() <= 1e-9 * np.abs(wk[1]).max()
    pairs, npairs = ctx.orb_match(lmd[0:1], lmc[0:1], lmd[1:2], lmc[1:2], 0.8)
    m = int(npairs[0])
    hp = pairs.cpu().numpy()[0, :m]
    assert np.array_equal(hp, np.array(O.orb_match(wa[2], wb[2], 0.8))) and m >= 60
    # gather on the device: cv::Point3f(kf0->lm_3d[queryIdx]), cv::Point2f(kf1->lm_2d[trainIdx])   (:643-652)
    cap = 1024
    qi, ti = pairs[0, :m, 0].long(), pairs[0, :m, 1].long()
    d3 = torch.zeros((1, cap, 3), dtype=torch.float32, device="cuda")
    d2 = torch.zeros((1, cap, 2), dtype=torch.float32, device="cuda")
    d3[0, :m] = lm3[0, qi].float()
    d2[0, :m] = lm2[1, ti]
    pose, mask, ninl = ctx.pnp_ransac(d3, d2, torch.tensor([m], dtype=torch.int32, device="cuda"), K4, [11])
    n_want, pose_want, mask_want = O.solve_pnp_ransac(h3[0, hp[:, 0]], h2[1, hp[:, 1]], K4, iterative=False, iterations=100, reproj=2.0,
                                                      conf=0.99, seed=11)
    n_inl = int(ninl[0])
    assert n_inl == n_want and np.array_equal(pose.cpu().numpy()[0], pose_want) and np.array_equal(mask.cpu().numpy()[0, :m], mask_want)
    assert n_inl >= 20 and n_inl / m >= 0.5                            # acceptance rule (:677-686, ratioRansac 0.5, minPts 20)
    Ra, tta = tr.T_c_w(ta)
    Rb, ttb = tr.T_c_w(tb)
    R_gt = Rb @ Ra.T
    R, t = G.pose7_to_Rt(pose.cpu().numpy()[0])
    assert np.degrees(np.arccos(np.clip((np.trace(R @ R_gt.T) - 1) / 2, -1, 1))) < 0.6 and np.linalg.norm(t - (ttb - R_gt @ tta)) < 0.05   # (EPnP on the inliers, unrefined: what SOLVEPNP_P3P ends with)


def test_lc_keyframe_landmarks_parity(ctx):
    """STEP 1.5 / 1.6 of the loop-closing keyframe (vo_loopclosing.cpp:255-372): ORB keypoints -> stereo LK + DLT (or the depth image)
    -> the lists without the keypoints that got no position.  Kept set, order, 2-D and descriptors exact; 3-D as the tracker's DLT."""
    import os
    import tempfile
    import torch
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_lckf_gpu.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    trs = [synth.Trajectory(5), synth.Trajectory(9)]
    rnd = synth.Renderer("cuda")
    i0, i1 = rnd.stereo_frame(trs, 1.0, 20)
    kps, desc, cnt, _ = ctx.orb_detect_and_compute(i0, cap=1024)
    hk, hd, hc = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy()
    # (a) rectified stereo, two keyframes per call
    lm2, lm3, lmd, lmc = [t.cpu().numpy() for t in ctx.lc_keyframe_landmarks(i0, i1, 0, kps, desc, cnt, P0=P0, P1=P1)]
    for s in range(2):
        w2, w3, wd = O.lc_keyframe_landmarks(i0[s].cpu().numpy(), i1[s].cpu().numpy(), 0, hk[s, :hc[s]], hd[s, :hc[s]], P0, P1)
        assert lmc[s] == len(w2) and 100 < len(w2) <= hc[s], (s, lmc[s], len(w2), hc[s])
        assert np.array_equal(lm2[s, :lmc[s]], w2) and np.array_equal(lmd[s, :lmc[s]], wd)
        assert np.abs(lm3[s, :lmc[s]] - w3).max() <= 1e-9 * max(1.0, np.abs(w3).max()), np.abs(lm3[s, :lmc[s]] - w3).max()
    # (b) the depth camera: Z16 image, whole metres as the reference's integer division leaves them
    j0, d16 = rnd.depth_frame(trs, 1.0, 20)
    hz = d16.cpu().numpy().view(np.uint16)
    m2, m3, md, mc = [t.cpu().numpy() for t in ctx.lc_keyframe_landmarks(None, d16, 2, kps, desc, cnt, K4=K4)]
    for s in range(2):
        w2, w3, wd = O.lc_keyframe_landmarks(None, hz[s], 2, hk[s, :hc[s]], hd[s, :hc[s]], K4=K4)
        assert mc[s] == len(w2) and len(w2) > 50
        assert np.array_equal(m2[s, :mc[s]], w2) and np.array_equal(md[s, :mc[s]], wd) and np.array_equal(m3[s, :mc[s]], w3)
        assert set(np.unique(w3[:, 2])) <= set(float(v) for v in range(1, 11))
    # (c) unrectified stereo: the reference's case is empty
    assert ctx.lc_keyframe_landmarks(i0, i1, 1, kps, desc, cnt)[3].cpu().numpy().tolist() == [0, 0]
    # (d) descriptors compacted in place
    d2 = desc.clone()
    q2, q3, qd, qc = ctx.lc_keyframe_landmarks(i0, i1, 0, kps, d2, cnt, P0=P0, P1=P1, in_place=True)
    assert qd.data_ptr() == d2.data_ptr() and np.array_equal(qc.cpu().numpy(), lmc)
    for s in range(2):
        assert np.array_equal(d2[s, :lmc[s]].cpu().numpy(), lmd[s, :lmc[s]])
    # (e) the kept lists feed solvePnPRansac directly (float positions, as cv::Point3f in isLoopClosureKF :643-650)
    pose, mask, ninl = ctx.pnp_ransac(torch.from_numpy(lm3.astype(np.float32)).cuda(), torch.from_numpy(lm2).cuda(),
                                      torch.from_numpy(lmc).cuda(), K4, [3, 4])
    assert int(ninl.min()) > 0.8 * int(lmc.min())                        # a keyframe against itself: identity pose
    hp = pose.cpu().numpy()
    assert np.abs(hp[:, :3]).max() < 0.02 and np.abs(np.abs(hp[:, 6]) - 1).max() < 1e-3, hp


def test_loop_entry_points_reject_bad_input(ctx):
    """capacities and argument checks of the loop-closing entry points fail loudly (error code + message), nothing is launched"""
    import ctypes as C
    import torch
    import flvis_amd
    lib, h = ctx._lib, ctx._h
    cp = np.array([0, 2, 2, 2], np.int32)
    ci = np.array([1, 2], np.int32)
    ds = np.zeros((3, 32), np.uint8)
    wt = np.array([0.0, 1.0, 1.0])
    wi = np.array([-1, 0, 1], np.int32)
    ctx.bow_set_vocabulary(cp, ci, ds, wt, wi)                                   # a valid two-word tree
    with pytest.raises(flvis_amd.FlvisError):
        ctx.bow_set_vocabulary(np.array([1, 2, 2, 2], np.int32), ci, ds, wt, wi)  # node 0 is not the root
    with pytest.raises(flvis_amd.FlvisError):
        ctx.bow_set_vocabulary(cp, np.array([1, 7], np.int32), ds, wt, wi)       # child index out of range
    # links that do not form a tree (k_bow_words walks down until it meets a leaf: such a table used to hang the GPU)
    ds5, wt5, wi5 = np.zeros((5, 32), np.uint8), np.array([0.0, 0.0, 1.0, 1.0, 1.0]), np.array([-1, -1, 0, 1, 2], np.int32)
    ctx.bow_set_vocabulary(np.array([0, 2, 4, 4, 4, 4], np.int32), np.array([1, 2, 3, 4], np.int32), ds5, wt5, wi5)   # valid: 0 -> {1, 2}, 1 -> {3, 4}
    with pytest.raises(flvis_amd.FlvisError):      # node 1 lists itself as its child (and node 4 hangs nowhere)
        ctx.bow_set_vocabulary(np.array([0, 2, 4, 4, 4, 4], np.int32), np.array([1, 2, 1, 3], np.int32), ds5, wt5, wi5)
    with pytest.raises(flvis_amd.FlvisError):      # node 3 is the child of two nodes
        ctx.bow_set_vocabulary(np.array([0, 2, 4, 4, 4, 4], np.int32), np.array([1, 3, 3, 4], np.int32), ds5, wt5, wi5)
    with pytest.raises(flvis_amd.FlvisError):      # nodes 3 and 4 form a cycle that the root does not reach
        ctx.bow_set_vocabulary(np.array([0, 2, 2, 2, 3, 4], np.int32), np.array([1, 2, 4, 3], np.int32), ds5,
                               np.array([0.0, 1.0, 1.0, 0.0, 0.0]), np.array([-1, 0, 1, -1, -1], np.int32))
    ctx.bow_set_vocabulary(cp, ci, ds, wt, wi)
    big = torch.zeros((1, 4096, 32), dtype=torch.uint8, device="cuda")
    with pytest.raises(flvis_amd.FlvisError):
        ctx.bow_transform(big, torch.zeros(1, dtype=torch.int32, device="cuda"))   # more than 2048 descriptors per keyframe
    p3 = torch.zeros((1, 2048, 3), dtype=torch.float32, device="cuda")
    p2 = torch.zeros((1, 2048, 2), dtype=torch.float32, device="cuda")
    with pytest.raises(flvis_amd.FlvisError):
        ctx.pnp_ransac(p3, p2, torch.zeros(1, dtype=torch.int32, device="cuda"), [1.0, 1.0, 0.0, 0.0], [1])   # > 1024 correspondences
    kp = torch.zeros((1, 64, 6), dtype=torch.float32, device="cuda")
    dd = torch.zeros((1, 64, 32), dtype=torch.uint8, device="cuda")
    c1 = torch.zeros(1, dtype=torch.int32, device="cuda")
    im = torch.zeros((1, 64, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(flvis_amd.FlvisError) as e:
        ctx.lc_keyframe_landmarks(im, im, 0, kp, dd, c1)                          # stereo without projection matrices
    assert "P0 and P1" in str(e.value)
    with pytest.raises(flvis_amd.FlvisError):
        ctx.lc_keyframe_landmarks(im, im, 3, kp, dd, c1)                          # no such camera type
    with pytest.raises(flvis_amd.FlvisError):
        ctx.lc_keyframe_landmarks(None, im, 2, kp, dd, c1)                        # depth camera without intrinsics
    with pytest.raises(flvis_amd.FlvisError):
        ctx.lc_keyframe_landmarks(im, im, 1, torch.zeros((1, 4096, 6), dtype=torch.float32, device="cuda"),
                                  torch.zeros((1, 4096, 32), dtype=torch.uint8, device="cuda"), c1)   # > 2048 keypoints
