"""Synthetic sliding-window BA problems (pattern of g2o examples/ba/ba_demo.cpp: points, cameras, pixel noise, outliers)."""
import numpy as np

import _geom as G

K4 = np.array([384.16455078125, 384.16455078125, 320.2144470214844, 238.94403076171875])


def make_sequence(seed, n_kf=14, n_lm=300, pix_sigma=0.5, outlier_frac=0.05, lm_sigma=0.05, pose_sigma=(0.02, 0.0087)):
    """Returns ground truth and the noisy keyframe stream a tracker would publish (KeyFrame.msg payloads)."""
    rng = np.random.default_rng(seed)
    Pw = np.stack([rng.uniform(-3, 3, n_lm), rng.uniform(-2, 2, n_lm), rng.uniform(2, 6, n_lm)], 1)
    kfs = []
    gt = []
    for k in range(n_kf):
        tc = np.array([0.12 * k, 0.03 * np.sin(0.7 * k), 0.02 * k])          # camera centre in world
        R = G.rodrigues(np.array([0.01 * np.sin(k), 0.02 * k - 0.1, 0.005 * k]))  # R_c_w
        t = -R @ tc
        gt.append((R, t))
        uv = G.project(R, t, Pw, K4)
        Xc = Pw @ R.T + t
        vis = (uv[:, 0] > 5) & (uv[:, 0] < 635) & (uv[:, 1] > 5) & (uv[:, 1] < 475) & (Xc[:, 2] > 0.5)
        vis &= rng.random(n_lm) < 0.8
        idx = np.nonzero(vis)[0]
        z = uv[idx] + rng.normal(0, pix_sigma, (len(idx), 2))
        out = rng.random(len(idx)) < outlier_frac
        z[out] = np.stack([rng.uniform(0, 640, out.sum()), rng.uniform(0, 480, out.sum())], 1)
        lm3 = Pw[idx] + rng.normal(0, lm_sigma, (len(idx), 3))
        Rn = G.rodrigues(rng.normal(0, pose_sigma[1], 3)) @ R
        tn = t + rng.normal(0, pose_sigma[0], 3)
        kfs.append(dict(frame_id=10 + 3 * k, pose7=G.pose7(Rn, tn), lm_id=(idx + 100).astype(np.int64), lm_2d=z,
                        lm_3d=lm3, outlier=out))
    return dict(Pw=Pw, gt=gt, kfs=kfs)
