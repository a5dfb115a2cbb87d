"""CPU test of the keyframe-rate front half of the reference's loop closing, assembled from the oracle's pieces exactly as
src/backend/vo_loopclosing.cpp does it: ORB on img0 of both keyframes (:242-243), stereo LK (5 levels) +
stereo triangulation for the 3-D of keyframe 0's ORB points (:283-306), BFMatcher knn x2 + mutual/ratio test (:601-639),
solvePnPRansac(P3P, 100 iterations, 2.0 px, 0.99) (:670), the acceptance rule (:677-686).  Two views of the synthetic room
taken 0.6 s apart stand in for a revisit: the recovered relative pose must match the ground truth."""
import os
import tempfile

import numpy as np

import _geom as G
import _oracle as O


def test_orb_match_pnp_recovers_the_relative_pose_of_two_keyframes():
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_loop.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    P0, P1 = np.array(list(cfg.P0)), np.array(list(cfg.P1))
    tr = synth.Trajectory(5)
    rnd = synth.Renderer("cpu")
    ta, tb = 1.0, 1.6
    a0, a1 = [x[0].numpy() for x in rnd.stereo_frame([tr], ta, 20)]
    b0, _ = [x[0].numpy() for x in rnd.stereo_frame([tr], tb, 32)]
    # keyframe 0: ORB + stereo depth (the reference tracks the ORB points into img1 with 5-level LK from the same pixel)
    ka, da = O.orb_detect_and_compute(a0)
    pts = ka[:, :2].copy()
    nxt, st = O.lk(a0, a1, pts, pts, max_level=5)
    p3d = np.zeros((len(ka), 3))
    has3d = np.zeros(len(ka), bool)
    for i in range(len(ka)):
        if st[i] == 1:
            pc = O.triangulate_dlt(pts[i].astype(np.float64), nxt[i].astype(np.float64), P0, P1)
            if 0 < pc[2] < 20:               # Triangulation::trignaulationPtFromStereo's validity range
                p3d[i], has3d[i] = pc, True
    assert has3d.sum() > 0.7 * len(ka)
    # keyframe 1: ORB only
    kb, db = O.orb_detect_and_compute(b0)
    pairs = O.orb_match(da, db, 0.8)
    pairs = np.array([pr for pr in pairs if has3d[pr[0]]])
    assert len(pairs) >= 60, len(pairs)
    n_inl, pose, mask = O.solve_pnp_ransac(p3d[pairs[:, 0]], kb[pairs[:, 1], :2], K4, iterative=False, iterations=100,
                                           reproj=2.0, conf=0.99, seed=11)
    # acceptance rule of the reference (lcKF parameters of the d435i yaml: ratioRansac 0.5, minPts 20)
    assert n_inl >= 20 and n_inl / len(pairs) >= 0.5, (n_inl, len(pairs))
    # ground truth: T_c1_c0 = T_c1_w * T_w_c0
    Ra, tta = tr.T_c_w(ta)
    Rb, ttb = tr.T_c_w(tb)
    R_gt = Rb @ Ra.T
    t_gt = ttb - R_gt @ tta
    R, t = G.pose7_to_Rt(pose)
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ R_gt.T) - 1) / 2, -1, 1)))
    # (SOLVEPNP_P3P ends with EPnP on the RANSAC inliers, unrefined; the 3-D points carry the stereo triangulation's error)
    assert ang < 1.0 and np.linalg.norm(t - t_gt) < 0.08, (ang, np.linalg.norm(t - t_gt))
    # the mutual + ratio test leaves almost only geometrically consistent pairs
    proj = (p3d[pairs[:, 0]] @ R_gt.T + t_gt)
    uv = np.stack([K4[0] * proj[:, 0] / proj[:, 2] + K4[2], K4[1] * proj[:, 1] / proj[:, 2] + K4[3]], 1)
    err = np.linalg.norm(uv - kb[pairs[:, 1], :2], axis=1)
    assert np.mean(err < 3.0) > 0.8
