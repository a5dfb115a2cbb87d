"""The optional IMU rotation factor of the window BA (north_star: "reprojection + IMU-preintegration factors"; the reference has
no such edge, so this is an addition behind a flag, off by default): residual r = Log(dq^T R_b(a)^T R_b(b)) between two
consecutive keyframes with dq the gyro preintegration.  Oracle side: analytic Jacobians against central differences (pattern of
g2o's base_binary_edge.hpp:144-212), zero residual for a consistent measurement, and the factor's effect on a window."""
import ctypes as C

import numpy as np

import _ba_synth as B
import _geom as G
import _oracle as O


def _d(a):
    a = np.ascontiguousarray(a, np.float64)
    return a.ctypes.data_as(C.POINTER(C.c_double)), a


def _rand_pose(rng):
    R = G.rodrigues(rng.normal(0, 0.4, 3))
    return G.pose7(R, rng.normal(0, 1.0, 3)), R


def _quat_wxyz(R):
    p7 = G.pose7(R, np.zeros(3))
    return np.array([p7[6], p7[3], p7[4], p7[5]])


def _lin(Ta, Tb, qcb, dq):
    r, Ja, Jb = np.zeros(3), np.zeros(9), np.zeros(9)
    pa, _a = _d(Ta); pb, _b = _d(Tb); pq, _q = _d(qcb); pd, _dd = _d(dq)
    O.lib().ref_imu_edge_linearize(pa, pb, pq, pd, r.ctypes.data_as(C.POINTER(C.c_double)), Ja.ctypes.data_as(C.POINTER(C.c_double)),
                                   Jb.ctypes.data_as(C.POINTER(C.c_double)))
    return r, Ja.reshape(3, 3), Jb.reshape(3, 3)


def _oplus(T, dx):
    out = np.zeros(7)
    pt, _t = _d(T); px, _x = _d(dx)
    O.lib().ref_g2o_oplus(pt, px, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_imu_edge_jacobians_match_central_differences():
    rng = np.random.default_rng(3)
    for trial in range(6):
        Ta, Ra = _rand_pose(rng)
        Tb, Rb = _rand_pose(rng)
        Rcb = G.rodrigues(rng.normal(0, 0.8, 3))
        # measurement = truth perturbed by up to ~0.1 rad: the residual is not small
        Rwb_a, Rwb_b = Ra.T @ Rcb, Rb.T @ Rcb
        dR = Rwb_a.T @ Rwb_b @ G.rodrigues(rng.normal(0, 0.05, 3))
        qcb, dq = _quat_wxyz(Rcb), _quat_wxyz(dR)
        r, Ja, Jb = _lin(Ta, Tb, qcb, dq)
        assert np.linalg.norm(r) < 0.5
        h = 1e-6
        for k in range(6):
            dx = np.zeros(6); dx[k] = h
            ra_p, _, _ = _lin(_oplus(Ta, dx), Tb, qcb, dq)
            ra_m, _, _ = _lin(_oplus(Ta, -dx), Tb, qcb, dq)
            rb_p, _, _ = _lin(Ta, _oplus(Tb, dx), qcb, dq)
            rb_m, _, _ = _lin(Ta, _oplus(Tb, -dx), qcb, dq)
            na, nb = (ra_p - ra_m) / (2 * h), (rb_p - rb_m) / (2 * h)
            if k < 3:      # rotation part of the tangent (omega first in g2o's SE3Quat)
                assert np.allclose(Ja[:, k], na, atol=1e-6), (trial, k, Ja[:, k], na)
                assert np.allclose(Jb[:, k], nb, atol=1e-6), (trial, k, Jb[:, k], nb)
            else:          # translations do not enter a rotation residual
                assert np.abs(na).max() < 1e-9 and np.abs(nb).max() < 1e-9


def test_consistent_measurement_has_zero_residual():
    rng = np.random.default_rng(4)
    Ta, Ra = _rand_pose(rng)
    Tb, Rb = _rand_pose(rng)
    Rcb = G.rodrigues(np.array([0.3, -1.1, 0.5]))
    dR = (Ra.T @ Rcb).T @ (Rb.T @ Rcb)
    r, _, _ = _lin(Ta, Tb, _quat_wxyz(Rcb), _quat_wxyz(dR))
    assert np.abs(r).max() < 1e-12


def _run_window(seed, imu, sigma_g=0.002, rot_noise=0.0, pix_sigma=1.0):
    """14 keyframes through the oracle's LocalMap; the keyframe poses carry 0.5 deg rotation noise (as _ba_synth makes them).  With
    imu: every keyframe comes with the TRUE relative body rotation to its predecessor (+ rot_noise).  Returns, per optimisation,
    the mean error of the RELATIVE rotation between consecutive window keyframes (the gauge -- the fixed oldest pose -- drops out)."""
    seq = B.make_sequence(seed, n_kf=14, n_lm=120, outlier_frac=0.0, pix_sigma=pix_sigma)
    W = 8
    lm = O.LocalMap(W, B.K4)
    Rcb = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])   # camera <- body (inverse of the D435 mount)
    rng = np.random.default_rng(seed + 100)
    if imu:
        qcb, _q = _d(_quat_wxyz(Rcb))
        O.lib().ref_localmap_set_imu_factor(lm.h, 1, C.c_double(sigma_g), qcb)
    errs = []
    for k, kf in enumerate(seq["kfs"]):
        if imu and k > 0:
            Ra, Rb = seq["gt"][k - 1][0], seq["gt"][k][0]
            dR = (Ra.T @ Rcb).T @ (Rb.T @ Rcb) @ (G.rodrigues(rng.normal(0, rot_noise, 3)) if rot_noise > 0 else np.eye(3))
            pd, _dq = _d(_quat_wxyz(dR))
            O.lib().ref_localmap_next_imu(pd, C.c_double(0.15))
        out = lm.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        if out is None:
            continue
        poses, _, _ = lm.poses()             # slot j holds the newest keyframe kk <= k with kk % W == j (ring, vo_localmap.cpp:166-225)
        by_k = {k - ((k - j) % W): G.pose7_to_Rt(poses[j])[0] for j in range(W)}
        e = []
        for kk in sorted(by_k):
            if kk - 1 in by_k:
                rel = by_k[kk - 1] @ by_k[kk].T
                rel_gt = seq["gt"][kk - 1][0] @ seq["gt"][kk][0].T
                e.append(np.arccos(np.clip((np.trace(rel @ rel_gt.T) - 1) / 2, -1, 1)))
        errs.append(np.mean(e))
    return np.array(errs)


def test_imu_factor_is_off_by_default_and_tightens_rotation_when_on():
    base = _run_window(31, imu=False, pix_sigma=2.0)
    again = _run_window(31, imu=False, pix_sigma=2.0)
    assert np.array_equal(base, again)
    with_imu = _run_window(31, imu=True, pix_sigma=2.0)
    assert len(with_imu) == len(base) >= 5
    # exact relative rotations with a tight information (sigma_g = 0.002 rad/s/sqrt(Hz), dt = 0.15 s => sigma = 0.8 mrad): the
    # relative rotations inside the window follow the gyro instead of the 2 px reprojection noise
    assert with_imu.mean() < 0.5 * base.mean(), (with_imu.mean(), base.mean())


def test_tracker_preintegration_follows_the_true_body_rotation():
    """The oracle tracker's per-keyframe gyro preintegration (VIMOTION accumulators, restarted at every keyframe) against the
    analytic body rotation of the synthetic trajectory between the two keyframe times."""
    import os
    import tempfile
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_imufac.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    trk = O.Tracker(cfg, 7)
    tr = synth.Trajectory(5)
    rnd = synth.Renderer("cpu")
    t_prev, frame0, t_last_kf, checked = -0.05, None, None, 0
    for f in range(50 + 24):
        t = f / synth.FRAME_HZ
        for s in synth.imu_samples(tr, 5, t_prev, t):
            trk.imu(s[0], s[1:4], s[4:7])
        t_prev = t
        if f >= 50 or frame0 is None:
            i0, i1 = rnd.stereo_frame([tr], t, f)
            frame0 = (i0[0].numpy(), i1[0].numpy())
        r = trk.image(t, frame0[0], frame0[1])
        if not r["new_keyframe"]:
            continue
        valid, dq, dt = trk.keyframe_imu()
        if t_last_kf is None:
            assert not valid                       # the keyframe of init_frame() starts the chain
        else:
            assert valid and abs(dt - (t - t_last_kf)) < 1e-9
            dR = G.quat_to_R(np.array([dq[1], dq[2], dq[3], dq[0]]))
            dR_true = tr.R_w_i(t_last_kf).T @ tr.R_w_i(t)
            err = np.arccos(np.clip((np.trace(dR.T @ dR_true) - 1) / 2, -1, 1))
            # 200 Hz rectangle rule + 2 mrad/s bias and noise over <= 0.3 s
            assert err < 4e-3, (f, err)
            checked += 1
        t_last_kf = t
    assert checked >= 4
