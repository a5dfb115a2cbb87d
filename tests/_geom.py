"""numpy helpers for geometry tests: SE3 as (R, t), pose7 = [tx ty tz qx qy qz qw]."""
import numpy as np


def quat_to_R(q):  # q = (x, y, z, w)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()  # x y z w


def rodrigues(r):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(r).as_matrix()


def pose7(R, t):
    q = R_to_quat(R)
    if q[3] < 0:
        q = -q
    return np.concatenate([t, q])


def pose7_to_Rt(p):
    return quat_to_R(p[3:7]), np.array(p[:3], dtype=float)


def project(R, t, P, K4):
    X = P @ R.T + t
    return np.stack([K4[0] * X[:, 0] / X[:, 2] + K4[2], K4[1] * X[:, 1] / X[:, 2] + K4[3]], 1)


def random_scene(rng, n, K4, w=640, h=480, zmin=1.5, zmax=8.0):
    """n points visible from a camera at identity; returns them in that camera frame plus their pixels."""
    uv = np.stack([rng.uniform(20, w - 20, n), rng.uniform(20, h - 20, n)], 1)
    z = rng.uniform(zmin, zmax, n)
    P = np.stack([(uv[:, 0] - K4[2]) / K4[0] * z, (uv[:, 1] - K4[3]) / K4[1] * z, z], 1)
    return P, uv
