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
            # the position preintegration over the same interval against the analytic trajectory:
            # dp = R_a^T (p_b - p_a - v_a dt + 1/2 g_w dt^2) with the TRUE velocity (dp does not depend on the filter's)
            dp, va = trk.keyframe_imu_pos()
            h = 1e-4
            v_a = (tr.pos(t_last_kf + h) - tr.pos(t_last_kf - h)) / (2 * h)
            dp_true = tr.R_w_i(t_last_kf).T @ (tr.pos(t) - tr.pos(t_last_kf) - v_a * dt + 0.5 * np.array([0, 0, -9.81]) * dt * dt)
            # 0.05 m/s^2 accelerometer bias + 0.02 m/s^2 noise + the rectangle rule over <= 0.3 s
            assert np.abs(dp - dp_true).max() < 1.5e-2, (f, dp, dp_true)
            assert np.all(np.isfinite(va))
            checked += 1
        t_last_kf = t
    assert checked >= 4


# ---------------------------------------------------------------------------------------------------------------- position rows
def _lin_pos(Ta, Tb, qcb, tcb, dp, va, dt):
    r, Ja, Jb = np.zeros(3), np.zeros(18), np.zeros(18)
    args = [_d(x) for x in (Ta, Tb, qcb, tcb, dp, va)]
    P = C.POINTER(C.c_double)
    O.lib().ref_imu_edge_linearize_pos(args[0][0], args[1][0], args[2][0], args[3][0], args[4][0], args[5][0], C.c_double(dt),
                                       r.ctypes.data_as(P), Ja.ctypes.data_as(P), Jb.ctypes.data_as(P))
    return r, Ja.reshape(3, 6), Jb.reshape(3, 6)


def _body(T7, Rcb, tcb):
    """body attitude R_w_b and position p_b in the world from the camera pose T_c_w and the camera-from-body extrinsic"""
    R, t = G.pose7_to_Rt(T7)
    return R.T @ Rcb, R.T @ (tcb - t)


def test_imu_position_rows_jacobians_and_consistency():
    """r_p = R_b(a)^T (p_b(b) - p_b(a) - v_a dt + 1/2 g_w dt^2) - dp: zero for a displacement preintegrated from the truth (constant
    world acceleration, constant attitude: closed form), and the analytic 3 x 6 Jacobians with respect to g2o's left updates of both
    camera poses equal central differences -- both poses enter through rotation AND translation."""
    rng = np.random.default_rng(11)
    gw = np.array([0.0, 0.0, -9.81])
    for trial in range(6):
        Ta, Ra = _rand_pose(rng)
        Tb, Rb = _rand_pose(rng)
        Rcb = G.rodrigues(rng.normal(0, 0.8, 3))
        tcb = rng.normal(0, 0.1, 3)
        qcb = _quat_wxyz(Rcb)
        dt = 0.15 + 0.1 * trial
        Rwa, pa = _body(Ta, Rcb, tcb)
        _, pb = _body(Tb, Rcb, tcb)
        va = rng.normal(0, 0.5, 3)
        dp_true = Rwa.T @ (pb - pa - va * dt + 0.5 * gw * dt * dt)            # what an exact preintegration would have measured
        r0, _, _ = _lin_pos(Ta, Tb, qcb, tcb, dp_true, va, dt)
        assert np.abs(r0).max() < 1e-12
        dp = dp_true + rng.normal(0, 0.02, 3)                                 # a 2 cm residual
        r, Ja, Jb = _lin_pos(Ta, Tb, qcb, tcb, dp, va, dt)
        assert np.allclose(r, dp_true - dp, atol=1e-12)
        h = 1e-6
        for k in range(6):
            dx = np.zeros(6); dx[k] = h
            na = (_lin_pos(_oplus(Ta, dx), Tb, qcb, tcb, dp, va, dt)[0] - _lin_pos(_oplus(Ta, -dx), Tb, qcb, tcb, dp, va, dt)[0]) / (2 * h)
            nb = (_lin_pos(Ta, _oplus(Tb, dx), qcb, tcb, dp, va, dt)[0] - _lin_pos(Ta, _oplus(Tb, -dx), qcb, tcb, dp, va, dt)[0]) / (2 * h)
            assert np.allclose(Ja[:, k], na, atol=1e-6), (trial, k, Ja[:, k], na)
            assert np.allclose(Jb[:, k], nb, atol=1e-6), (trial, k, Jb[:, k], nb)
    # the preintegration sums themselves: constant specific force f and rate over n samples reproduce the closed form
    dtk, n = 0.005, 40
    f, w = np.array([0.3, -0.2, -9.6]), np.array([0.02, -0.01, 0.03])
    dR, dv, dp = np.eye(3), np.zeros(3), np.zeros(3)
    for _ in range(n):
        a = dR @ f
        dp = dp + dv * dtk + 0.5 * a * dtk * dtk
        dv = dv + a * dtk
        dR = dR @ G.rodrigues(w * dtk)
    T = n * dtk
    assert np.allclose(dp, 0.5 * f * T * T, atol=6e-4) and np.allclose(dv, f * T, atol=6e-3)   # small rotation: nearly the free-fall form


def _run_window_pos(seed, sigma_a, pos_noise=0.0):
    """_run_window with the full factor: true relative rotation AND true preintegrated displacement / velocity per keyframe.  Returns
    the mean error of the RELATIVE body translation between consecutive window keyframes per optimisation."""
    seq = B.make_sequence(seed, n_kf=14, n_lm=120, outlier_frac=0.0, pix_sigma=2.0)
    W = 8
    lm = O.LocalMap(W, B.K4)
    Rcb = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    tcb = np.array([0.02, -0.01, 0.03])
    gw = np.array([0.0, 0.0, -9.81])
    rng = np.random.default_rng(seed + 200)
    dt = 0.15
    if sigma_a is not None:
        qcb, _q = _d(_quat_wxyz(Rcb))
        O.lib().ref_localmap_set_imu_factor(lm.h, 1, C.c_double(0.002), qcb)
        if sigma_a > 0:
            pt, _t = _d(tcb)
            O.lib().ref_localmap_set_imu_factor_pos(lm.h, C.c_double(sigma_a), pt)

    def body(k):
        R, t = seq["gt"][k]
        return R.T @ Rcb, R.T @ (tcb - t)
    errs = []
    for k, kf in enumerate(seq["kfs"]):
        if sigma_a is not None and k > 0:
            Rwa, pa = body(k - 1)
            Rwb, pb = body(k)
            pd, _dq = _d(_quat_wxyz(Rwa.T @ Rwb))
            O.lib().ref_localmap_next_imu(pd, C.c_double(dt))
            va = (pb - pa) / dt                                            # any velocity is consistent as long as dp is formed with it
            dp = Rwa.T @ (pb - pa - va * dt + 0.5 * gw * dt * dt) + rng.normal(0, pos_noise, 3)
            p1, _1 = _d(dp); p2, _2 = _d(va)
            O.lib().ref_localmap_next_imu_pos(p1, p2)
        out = lm.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        if out is None:
            continue
        poses, _, _ = lm.poses()
        by_k = {k - ((k - j) % W): poses[j] for j in range(W)}
        e = []
        for kk in sorted(by_k):
            if kk - 1 in by_k:
                pa_e = _body(by_k[kk - 1], Rcb, tcb)[1]
                pb_e = _body(by_k[kk], Rcb, tcb)[1]
                e.append(np.linalg.norm((pb_e - pa_e) - (body(kk)[1] - body(kk - 1)[1])))
        errs.append(np.mean(e))
    return np.array(errs)


def test_imu_position_rows_tighten_the_relative_translation():
    rot_only = _run_window_pos(31, sigma_a=0.0)
    full = _run_window_pos(31, sigma_a=0.05)
    assert len(full) == len(rot_only) >= 5
    # exact displacements at sigma_a = 0.05 m/s^2/sqrt(Hz), dt = 0.15 s (sigma = 1.7 mm): the relative translations inside the window
    # follow the accelerometer instead of the 2 px reprojection noise
    assert full.mean() < 0.5 * rot_only.mean(), (full.mean(), rot_only.mean())
    # sigma_a <= 0 is the rotation-only factor of round 2, bit for bit
    assert np.array_equal(rot_only, _run_window_pos(31, sigma_a=-1.0))
