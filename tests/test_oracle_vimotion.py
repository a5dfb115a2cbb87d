"""CPU test of the IMU filter restatement: the oracle's VIMOTION (oracle/ref_tracking.cpp) against an
independent numpy restatement of src/processing/vi_motion.cpp:34-209 -- first-sample roll/pitch from gravity, the
Madgwick-style gradient step with gain 10*beta during initialisation and beta afterwards, quaternion / position /
velocity propagation, the `float` narrowing of the scalar in scalar_multi_q (src/utils/include/kinetic_math.h:123)."""
import os
import tempfile

import numpy as np

import _oracle as O


def rpy2R(r, p, y):
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def R2Q(R):
    """Eigen::Quaterniond(Matrix3d) (Geometry/Quaternion.h, quaternionbase_assign_impl): w x y z."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([w, (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[1 + i] = 0.5 * t
    t = 0.5 / t
    q[0] = (R[k, j] - R[j, k]) * t
    q[1 + j] = (R[j, i] + R[i, j]) * t
    q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def Q2R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(q1, q2):   # q1_multi_q2, kinetic_math.h:113-121 (w x y z)
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return np.array([w2 * w1 - x2 * x1 - y2 * y1 - z2 * z1, x2 * w1 + w2 * x1 + z2 * y1 - y2 * z1,
                     y2 * w1 - z2 * x1 + w2 * y1 + x2 * z1, z2 * w1 + y2 * x1 - x2 * y1 + w2 * z1])


def smul(a, q):     # scalar_multi_q(const float a, ...): the scalar is narrowed to float
    return float(np.float32(a)) * q


def madgwick_s(q, acc):
    n = np.linalg.norm(acc)
    ax, ay, az = acc / n
    qw, qx, qy, qz = q
    s = np.array([
        2 * qx * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qy * (ax - 2 * qw * qy + 2 * qx * qz),
        2 * qw * (ay + 2 * qw * qx + 2 * qy * qz) + 2 * qz * (ax - 2 * qw * qy + 2 * qx * qz) - 4 * qx * (-2 * qx * qx - 2 * qy * qy + az + 1),
        2 * qz * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qw * (ax - 2 * qw * qy + 2 * qx * qz) - 4 * qy * (-2 * qx * qx - 2 * qy * qy + az + 1),
        2 * qx * (ax - 2 * qw * qy + 2 * qx * qz) + 2 * qy * (ay + 2 * qw * qx + 2 * qy * qz)])
    return s * np.linalg.norm(s)     # `s *= s.norm()` as written in the reference (not a normalisation)


class NumpyVimotion:
    def __init__(self, beta, g=9.81):
        self.beta, self.g = beta, g
        self.states = []            # (t, q, p, v)
        self.first, self.initialized = True, False

    def feed(self, t, acc, gyro):
        acc, gyro = np.asarray(acc, float), np.asarray(gyro, float)
        if not self.initialized:
            q = np.array([1.0, 0, 0, 0])
            if self.first:
                if np.linalg.norm(acc) - self.g < 0.3:
                    q = R2Q(rpy2R(np.arctan2(-acc[1], -acc[2]), np.arctan2(acc[0], -acc[2]), 0.0))
                    self.states.append((t, q, np.zeros(3), np.zeros(3)))
                    self.first = False
                return q, np.zeros(3), np.zeros(3)
            tp, qp, _, _ = self.states[-1]
            qdot = smul(0.5, qmul(qp, np.array([0.0, *gyro])))
            if np.linalg.norm(acc) - self.g < 0.3:
                qdot = qdot - 10 * self.beta * madgwick_s(qp, acc)
            qn = qp + smul(t - tp, qdot)
            qn /= np.linalg.norm(qn)
            self.states.append((t, qn, np.zeros(3), np.zeros(3)))
            if len(self.states) > 30:
                self.initialized = True
            return np.array([1.0, 0, 0, 0]), np.zeros(3), np.zeros(3)   # outputs stay at their initial values here
        tp, qp, pp, vp = self.states[-1]
        dt = t - tp
        qdot = smul(0.5, qmul(qp, np.array([0.0, *gyro])))
        if np.linalg.norm(acc) - self.g < 0.3:
            qdot = qdot - self.beta * madgwick_s(qp, acc)
        qn = qp + smul(dt, qdot)
        qn /= np.linalg.norm(qn)
        pn = pp + vp * dt
        vn = vp + ((Q2R(qp) @ acc) - np.array([0, 0, -self.g])) * dt
        self.states.append((t, qn, pn, vn))
        return qn, pn, vn


def test_vimotion_initialisation_and_propagation_match_numpy():
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_vimotion.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    trk = O.Tracker(cfg, 1)
    ref = NumpyVimotion(cfg.vifusion_para[0])
    tr = synth.Trajectory(4)
    smp = synth.imu_samples(tr, 4, -0.005, 0.6)          # 120 samples at 200 Hz, noisy, in the FLVIS IMU frame
    assert len(smp) >= 110
    worst = 0.0
    for k, r in enumerate(smp):
        out = trk.imu(r[0], r[1:4], r[4:7])              # q (w x y z), pos, vel
        q, pos, vel = ref.feed(r[0], r[1:4], r[4:7])
        if k == 0 or k > 31:                             # the first sample and everything after initialisation are outputs
            got_q = np.asarray(out[:4])
            if np.dot(got_q, q) < 0:
                got_q = -got_q
            worst = max(worst, np.abs(got_q - q).max(), np.abs(out[4:7] - pos).max(), np.abs(out[7:10] - vel).max())
    assert worst < 1e-12, worst
    assert np.linalg.norm(ref.states[-1][3]) > 1e-3      # the filter did integrate some motion


# ------------------------------------------------------------------------------------------------ vision coupling
def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qham(a, b):      # Hamilton product a*b (Eigen operator*), w x y z
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def R2rpy(R):
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], np.sqrt(R[2, 1] ** 2 + R[2, 2] ** 2)),
                     np.arctan2(R[1, 0], R[0, 0])])


def T_of(q, t):
    T = np.eye(4)
    T[:3, :3] = Q2R(np.asarray(q) / np.linalg.norm(q))
    T[:3, 3] = t
    return T


def pose7_of(T):
    q = R2Q(T[:3, :3])
    return np.array([T[0, 3], T[1, 3], T[2, 3], q[1], q[2], q[3], q[0]])


def T_of_pose7(p):
    return T_of([p[6], p[3], p[4], p[5]], p[:3])


class NumpyVimotionFull(NumpyVimotion):
    """+ viVisiontrigger (:117-137), viFindStateIdx (:347-383), viGetCorrFrameState (:416-435), viVisionRPCompensation
    (:437-464), viCorrectionFromVision (:212-342) with its quirks (gyro clamp tested on the ACC norm, (1-para_3) on both)."""

    def __init__(self, T_i_c, paras, g=9.81):
        super().__init__(paras[0], g)
        self.T_i_c, self.T_c_i = T_i_c, np.linalg.inv(T_i_c)
        self.p2, self.p3, self.p4 = paras[1], paras[2], paras[3]
        self.ba_sat, self.bw_sat = 0.5, 0.1                   # para5/6 are not forwarded (f2f_tracking.cpp:18-19)
        self.acc_bias, self.gyro_bias = np.zeros(3), np.zeros(3)

    def feed(self, t, acc, gyro):
        return super().feed(t, np.asarray(acc, float) - self.acc_bias, np.asarray(gyro, float) - self.gyro_bias)

    def vision_trigger(self):
        t, q, _, _ = self.states[-1]
        rpy = R2rpy(Q2R(q))
        q = R2Q(rpy2R(rpy[0], rpy[1], 0.0))
        q /= np.linalg.norm(q)
        self.states = [(t, q, np.zeros(3), np.zeros(3))]
        return q

    def find_idx(self, time):
        idx = 9999
        for i in range(len(self.states) - 1, -1, -1):
            idx = i
            if not (self.states[i][0] - time) > 0:
                break
        return idx if (idx > 0 and idx != 9999) else None

    def get_corr_frame_state(self, time):
        i = self.find_idx(time)
        if i is None:
            return None
        return np.linalg.inv(T_of(self.states[i][1], self.states[i][2]) @ self.T_i_c)

    def rp_compensation(self, time, T_c_w):
        T_w_i = np.linalg.inv(T_c_w) @ self.T_c_i
        before = R2rpy(T_w_i[:3, :3])
        i = self.find_idx(time)
        if i is None:
            return T_c_w
        imu = R2rpy(Q2R(self.states[i][1] / np.linalg.norm(self.states[i][1])))
        vim = np.array([imu[0], imu[1], before[2]])
        after = before * (1 - self.p2) + vim * self.p2
        T_after = np.eye(4)
        T_after[:3, :3] = rpy2R(*after)
        T_after[:3, 3] = T_w_i[:3, 3]
        return np.linalg.inv(T_after @ self.T_i_c)

    def correction(self, t_curr, Tcw_curr, t_last, Tcw_last):
        il, ic = self.find_idx(t_last), None
        if il is not None:
            ic = self.find_idx(t_curr)
        if il is None or ic is None or il == ic:
            return
        dt = t_curr - t_last
        im = il + (ic - il) // 2
        T_w_iA, T_w_iB = np.linalg.inv(Tcw_last) @ self.T_c_i, np.linalg.inv(Tcw_curr) @ self.T_c_i
        S = self.states
        T_w_ia, T_w_ib = T_of(S[il][1], S[il][2]), T_of(S[ic][1], S[ic][2])
        q_BA = R2Q((np.linalg.inv(T_w_iB) @ T_w_iA)[:3, :3])
        q_ba = R2Q((np.linalg.inv(T_w_ib) @ T_w_ia)[:3, :3])
        q_Bb = qham(q_BA, qconj(q_ba) / np.dot(q_ba, q_ba))
        if q_Bb[0] < 0 and False:
            q_Bb = -q_Bb
        gyro_est = q_Bb[1:] / dt
        vel_imu = np.mean([S[i][3] for i in range(il, ic + 1)], axis=0)
        vel_vis = (T_w_iB[:3, 3] - T_w_iA[:3, 3]) / dt
        dvel = vel_vis - vel_imu
        qm = S[im][1] / np.linalg.norm(S[im][1])
        acc_est = -(Q2R(qconj(qm)) @ dvel) / dt
        T_diff = T_w_iB @ np.linalg.inv(T_w_ib)
        for i in range(ic, len(S)):
            Tn = T_diff @ T_of(S[i][1], S[i][2])
            S[i] = (S[i][0], R2Q(Tn[:3, :3]), Tn[:3, 3].copy(), S[i][3] + dvel)
        na = np.linalg.norm(acc_est)
        if na > self.ba_sat:
            acc_est = acc_est * (self.ba_sat / na)
        if na > self.bw_sat:                                   # quirk: the ACC norm gates the gyro clamp
            gyro_est = gyro_est * (self.bw_sat / np.linalg.norm(gyro_est))
        if dt < 0.1:
            self.acc_bias = (1 - self.p3) * self.acc_bias + self.p3 * acc_est
            self.gyro_bias = (1 - self.p3) * self.gyro_bias + self.p4 * gyro_est


def _same_rot(qa, qb):
    qa, qb = np.asarray(qa) / np.linalg.norm(qa), np.asarray(qb) / np.linalg.norm(qb)
    return min(np.abs(qa - qb).max(), np.abs(qa + qb).max())


def test_vimotion_vision_coupling_matches_numpy():
    import ctypes as C
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_test_vimotion.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = O.load_config(p)
    T_i_c = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    paras = np.array(list(cfg.vifusion_para))
    lib = O.lib()
    lib.ref_vi_create.restype = C.c_void_p
    lib.ref_vi_create.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
    dp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    h = C.c_void_p(lib.ref_vi_create(dp(pose7_of(T_i_c)), 9.81, dp(paras)))
    ref = NumpyVimotionFull(T_i_c, paras)
    tr = synth.Trajectory(4)
    smp = synth.imu_samples(tr, 4, -0.005, 1.2)
    frame_dt = 1.0 / synth.FRAME_HZ

    def feed(r):
        out = np.zeros(10)
        lib.ref_vi_feed(h, C.c_double(r[0]), dp(r[1:4]), dp(r[4:7]), out.ctypes.data_as(C.POINTER(C.c_double)))
        ref.feed(r[0], r[1:4], r[4:7])

    def states():
        rows, b = np.zeros((400, 11)), np.zeros(6)
        n = lib.ref_vi_states(h, 400, rows.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)))
        return rows[:n], b

    def compare(tag):
        rows, b = states()
        assert len(rows) == len(ref.states), tag
        for r, (t, q, pos, vel) in zip(rows, ref.states):
            assert r[0] == t and _same_rot(r[1:5], q) < 1e-10, tag
            assert np.abs(r[5:8] - pos).max() < 1e-10 and np.abs(r[8:11] - vel).max() < 1e-10, tag
        assert np.abs(b[:3] - ref.acc_bias).max() < 1e-12 and np.abs(b[3:] - ref.gyro_bias).max() < 1e-12, tag

    k = 0
    while smp[k][0] <= 0.30:
        feed(smp[k])
        k += 1
    compare("after initialisation")
    q = np.zeros(4)
    lib.ref_vi_vision_trigger(h, q.ctypes.data_as(C.POINTER(C.c_double)))      # init_frame: yaw reset, queue cleared
    assert _same_rot(q, ref.vision_trigger()) < 1e-12
    t_frames = [0.35, 0.40, 0.45, 0.50, 0.55]
    T_prev = None
    for tf in t_frames:
        while k < len(smp) and smp[k][0] <= tf:
            feed(smp[k])
            k += 1
        # the IMU prior the tracker would use for this frame
        pose = np.zeros(7)
        ok = lib.ref_vi_get_corr_frame_state(h, C.c_double(tf), pose.ctypes.data_as(C.POINTER(C.c_double)))
        want = ref.get_corr_frame_state(tf)
        assert bool(ok) == (want is not None)
        if want is not None:
            assert np.abs(T_of_pose7(pose) - want).max() < 1e-10
        # a "vision" pose: the prior nudged by a few mm / mrad, then roll-pitch compensated, then fed back as correction
        T_vis = want.copy() if want is not None else np.linalg.inv(T_i_c)
        T_vis[:3, 3] += [0.004, -0.003, 0.002]
        T_vis[:3, :3] = T_vis[:3, :3] @ rpy2R(0.003, -0.002, 0.004)
        p7 = pose7_of(T_vis)
        lib.ref_vi_rp_compensation(h, C.c_double(tf), p7.ctypes.data_as(C.POINTER(C.c_double)))
        T_comp = ref.rp_compensation(tf, T_vis)
        assert np.abs(T_of_pose7(p7) - T_comp).max() < 1e-10
        if T_prev is not None:
            lib.ref_vi_correction(h, C.c_double(tf), dp(pose7_of(T_comp)), C.c_double(tf - frame_dt), dp(pose7_of(T_prev)))
            ref.correction(tf, T_comp, tf - frame_dt, T_prev)
            compare("after the correction at %.2f" % tf)
        T_prev = T_comp
    assert np.linalg.norm(ref.acc_bias) > 0 and np.linalg.norm(ref.gyro_bias) > 0       # the feedback did act
    lib.ref_vi_destroy(h)
