"""Synthetic loop-closure problems for the pose-graph optimisation (pattern of g2o's examples/sphere: a trajectory that returns to
its start, odometry with drift, one verified loop)."""
import numpy as np

import _geom as G


def inv7(p):
    R, t = G.pose7_to_Rt(p)
    return G.pose7(R.T, -R.T @ t)


def mul7(a, b):
    Ra, ta = G.pose7_to_Rt(a)
    Rb, tb = G.pose7_to_Rt(b)
    return G.pose7(Ra @ Rb, ta + Ra @ tb)


def make_loop(seed, n_kf=70, drift=(0.02, 0.004), loop_noise=(0.002, 0.001), extra_loops=0, radius=3.0):
    """Keyframes on a circle (the camera returns to its start).  Returns ground-truth T_c_w, drifted T_c_w (what the tracker
    produced), the loop list [(earlier, later)] and the verified relative poses T_later_earlier (what isLoopClosureKF yields)."""
    rng = np.random.default_rng(seed)
    gt = []
    for k in range(n_kf):
        a = 2 * np.pi * k / (n_kf - 4)                      # slightly more than one revolution
        c = np.array([radius * np.cos(a), radius * np.sin(a), 0.2 * np.sin(3 * a)])
        R_wc = G.rodrigues(np.array([0.0, 0.0, a + np.pi / 2])) @ G.rodrigues(np.array([0.1 * np.sin(a), 0.05, 0.0]))
        gt.append(G.pose7(R_wc.T, -R_wc.T @ c))            # T_c_w
    # odometry = true relative motion + noise, accumulated
    est = [gt[0].copy()]
    for k in range(1, n_kf):
        rel = mul7(gt[k], inv7(gt[k - 1]))                  # T_k_(k-1)
        R, t = G.pose7_to_Rt(rel)
        R = G.rodrigues(rng.normal(0, drift[1], 3)) @ R
        t = t + rng.normal(0, drift[0], 3)
        est.append(mul7(G.pose7(R, t), est[-1]))
    loops = [(2, n_kf - 1)]
    for e in range(extra_loops):
        loops.append((6 + 5 * e, n_kf - 8 - 3 * e))
    loop_poses = []
    for a, b in loops:
        rel = mul7(gt[b], inv7(gt[a]))                      # T_b_a (se_ji: from the earlier keyframe's camera to the later one's)
        R, t = G.pose7_to_Rt(rel)
        loop_poses.append(G.pose7(G.rodrigues(rng.normal(0, loop_noise[1], 3)) @ R, t + rng.normal(0, loop_noise[0], 3)))
    return dict(gt=np.array(gt), est=np.array(est), loops=np.array(loops, np.int32), loop_poses=np.array(loop_poses))


def centre_error(T_c_w, gt, idx):
    """camera-centre distance between estimate and ground truth after aligning at keyframe idx[0]"""
    def centre(p):
        R, t = G.pose7_to_Rt(p)
        return -R.T @ t
    return np.array([np.linalg.norm(centre(T_c_w[i]) - centre(gt[i])) for i in idx])


def loop_gap(T_c_w, gt, a, b):
    """how far the relative pose between keyframes a and b is from the true one: (translation [m], rotation [rad])"""
    rel, rel_gt = mul7(T_c_w[b], inv7(T_c_w[a])), mul7(gt[b], inv7(gt[a]))
    d = mul7(rel, inv7(rel_gt))
    R, t = G.pose7_to_Rt(d)
    return np.linalg.norm(t), np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
