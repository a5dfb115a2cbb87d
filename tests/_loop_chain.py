"""Test helper: (a) a synthetic sequence that closes a loop -- a camera that circles through the rendered room of flvis_amd.synth and
comes back to where it started, keyframes along the way, odometry poses with accumulated drift; (b) the control flow of the
reference's loop-closing nodelet around the per-keyframe features, assembled from the CPU oracle's functions:

    kfmsgProcess   src/backend/vo_loopclosing.cpp:191-391   T_c_w = T_c_w_odom * T_odom_map, keyframe appended
    pgoProcess     :393-518   similarity row, `size < 50` gate, isLoopCandidate, isLoopClosureKF, loop list, the PGO trigger
    isLoopClosureKF :593-735  mutual / ratio matches -> (3-D of the earlier keyframe, pixel of the later) -> solvePnPRansac ->
                              inlier ratio / count, |t| < 3, |log R| < 1.5
    loopClosureOnCovGraphG2ONew :742-944  (oracle/ref_pgo.cpp) and T_odom_map = T_odom_map * Tw1_w2

The nodelet's pgoProcess THREAD looks at whatever keyframe is newest whenever it comes round (a keyframe can be looked at twice or
never); here -- and in the product's flvis_loop_closer -- every keyframe is processed exactly once, in order."""
import math

import numpy as np

import _geom as G
import _oracle as O
import _pgo_synth as PS
from test_oracle_bow import ref_candidate, ref_score
from test_oracle_pgo import pgo as ref_pgo

LC_PARAMS = dict(lcKFStart=25, lcKFDist=18, lcKFMaxDist=50, lcKFLast=20, lcNKFClosest=2, ratioMax=0.5, ratioRansac=0.5, minPts=20,
                 minScore=0.12)     # launch/KITTI/KITTI.yaml:110-127


class LoopTrajectory:
    """body pose of a closed tour of the room: pos / R_w_i as flvis_amd.synth.Renderer reads them; period T seconds"""

    def __init__(self, T=60.0, phase=0.0):
        self.T, self.phase = T, phase

    def _a(self, t):
        return 2 * math.pi * t / self.T + self.phase

    def pos(self, t):
        a = self._a(t)
        return np.array([0.9 * math.cos(a) - 0.4, 1.1 * math.sin(a), 0.25 * math.sin(2 * a)])

    def R_w_i(self, t):
        from flvis_amd import synth
        a = self._a(t)
        return synth._rot_zyx(0.9 * math.sin(a), 0.12 * math.sin(2 * a + 0.3), 0.08 * math.cos(a))

    def acc(self, t):
        a, w = self._a(t), 2 * math.pi / self.T
        return np.array([-0.9 * w * w * math.cos(a), -1.1 * w * w * math.sin(a), -0.25 * 4 * w * w * math.sin(2 * a)])

    def omega_body(self, t, h=1e-4):
        """body rate from the central difference of R_w_i (the angles are slow sinusoids: error ~1e-9 rad/s)"""
        W = self.R_w_i(t).T @ (self.R_w_i(t + h) - self.R_w_i(t - h)) / (2 * h)
        return np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) / 2

    def T_c_w(self, t, rig):
        R_w_c = self.R_w_i(t) @ rig.R_i_c
        c = self.pos(t) + self.R_w_i(t) @ rig.t_i_c
        return R_w_c.T, -R_w_c.T @ c


def keyframe_times(n_kf, per_period):
    return [i * 60.0 / per_period for i in range(n_kf)]


def drifted_odometry(gt, seed, sigma_t=0.004, sigma_r=0.0015):
    """gt: list of T_c_w pose7; returns the poses an odometry with accumulating error would report"""
    rng = np.random.default_rng(seed)
    est = [gt[0].copy()]
    for k in range(1, len(gt)):
        rel = PS.mul7(gt[k], PS.inv7(gt[k - 1]))
        R, t = G.pose7_to_Rt(rel)
        est.append(PS.mul7(G.pose7(G.rodrigues(rng.normal(0, sigma_r, 3)) @ R, t + rng.normal(0, sigma_t, 3)), est[-1]))
    return est


def so3_log_norm(q):
    """|log R| of Sophus for a unit quaternion x y z w: the rotation angle in [0, pi]"""
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2])
    return 2.0 * math.atan2(n, abs(q[3]))


def pnp_seed(stream, kf_curr):
    return ((stream + 1) << 32) + kf_curr + 1


class RefLoopCloser:
    """one sequence.  add(feat, T_c_w_odom): feat = dict(bow=(ids, vals), lm2 [k,2] f32, lm3 [k,3] f64, lmd [k,32] u8)"""

    def __init__(self, K4, prm=LC_PARAMS, stream=0):
        self.K4, self.prm, self.stream = np.asarray(K4, np.float64), dict(prm), stream
        self.kfs, self.T_odom, self.T_c_w = [], [], []
        self.T_odom_map = np.array([0, 0, 0, 0, 0, 0, 1.0])
        self.loop_ids, self.loop_poses = [], []
        self.last_pgo = -5000
        self.rows = []

    def add(self, feat, T_c_w_odom):
        self.kfs.append(feat)
        self.T_odom.append(np.asarray(T_c_w_odom, np.float64).copy())
        self.T_c_w.append(PS.mul7(self.T_odom[-1], self.T_odom_map))          # :377
        return len(self.kfs) - 1

    def process(self):
        """the newest keyframe through pgoProcess; returns the event dict"""
        p = self.prm
        n = len(self.kfs)
        ev = dict(candidate=False, kf_prev=-1, kf_curr=n - 1, n_matches=0, n_inliers=0, accepted=False, optimised=False, pose=None)
        q = self.kfs[-1]["bow"]
        row = np.array([ref_score(q, kf["bow"]) for kf in self.kfs])
        self.rows.append(row)
        if n < 50:                                                              # :453
            return ev
        prev = ref_candidate(row, np.ones(n, np.uint8), p["lcKFDist"], p["lcKFMaxDist"], p["lcNKFClosest"], p["minScore"])
        if prev is None:
            return ev
        ev["candidate"], ev["kf_prev"] = True, prev
        k0, k1 = self.kfs[prev], self.kfs[-1]
        if len(k0["lmd"]) == 0 or len(k1["lmd"]) == 0:
            return ev
        pairs = np.array(O.orb_match(k0["lmd"], k1["lmd"], p["ratioMax"])).reshape(-1, 2)
        ev["n_matches"] = len(pairs)
        if len(pairs) < 5:                                                      # :666
            return ev
        p3d = k0["lm3"][pairs[:, 0]].astype(np.float32)
        p2d = k1["lm2"][pairs[:, 1]].astype(np.float32)
        ninl, pose, mask = O.solve_pnp_ransac(p3d, p2d, self.K4, iterative=False, iterations=100, reproj=2.0, conf=0.99,
                                              seed=pnp_seed(self.stream, n - 1))
        ev["n_inliers"], ev["pose"] = int(ninl), pose
        if ninl * 1.0 / len(pairs) < p["ratioRansac"] or ninl < p["minPts"]:    # :677
            return ev
        if not (np.linalg.norm(pose[:3]) < 3 and so3_log_norm(pose[3:7]) < 1.5):  # :686
            return ev
        ev["accepted"] = True
        self.loop_ids.append((prev, n - 1))
        self.loop_poses.append(pose.copy())
        thre = int((n / 100.0) * 2)                                             # :490
        if (n - 1) - self.last_pgo > thre:
            r, T, drift, stats = ref_pgo(np.array(self.T_c_w), np.ones(n, np.uint8), np.array(self.loop_ids, np.int32),
                                         np.array(self.loop_poses))
            if r:
                self.T_c_w = [t.copy() for t in T]
                self.T_odom_map = PS.mul7(self.T_odom_map, drift)               # :908
                ev["optimised"], ev["stats"] = True, stats
            self.last_pgo = n - 1
        return ev
