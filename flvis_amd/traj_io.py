"""Trajectory / dataset I/O and the ATE tool (harness code, CPU only).

Formats:
  * recorder format of the reference (src/independ_modules/vo_repub_rec.cpp:82-91): `stamp x y z qw qx qy qz`
    (what `evo_traj tum` is fed in results/1_readme, with qw first as the recorder writes it);
  * KITTI odometry poses (vo_repub_rec.cpp:100-111, bag/KITTI/dataset/poses.zip): 12 row-major entries of [R | t] per line;
  * EuRoC ASL ground truth (`state_groundtruth_estimate0/data.csv`): `#timestamp [ns], p_RS_R_x, p_RS_R_y, p_RS_R_z,
    q_RS_w, q_RS_x, q_RS_y, q_RS_z, ...`.
ATE = RMSE of the translation error after a least-squares rigid (optionally similarity) alignment (Umeyama 1991), on poses
associated by nearest timestamp -- the metric BASELINE.json quotes.
"""
import io
import zipfile

import numpy as np


def quat_to_rot(qw, qx, qy, qz):
    n = np.sqrt(qw * qw + qx * qx + qy * qy + qz * qz)
    qw, qx, qy, qz = qw / n, qx / n, qy / n, qz / n
    return np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                     [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                     [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])


def rot_to_quat(R):
    """-> (qw, qx, qy, qz), qw >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else -q


# ---------------------------------------------------------------------------------------------- recorder format
def write_stamped(path, stamps, positions, quats_wxyz):
    """`stamp x y z qw qx qy qz`, 6 significant digits like the reference's recorder (setprecision(6))."""
    with open(path, "w") as f:
        for t, p, q in zip(stamps, positions, quats_wxyz):
            f.write("%.9f %.6g %.6g %.6g %.6g %.6g %.6g %.6g\n" % (t, p[0], p[1], p[2], q[0], q[1], q[2], q[3]))


def read_stamped(path):
    """-> (stamps [n], positions [n,3], quats_wxyz [n,4])."""
    a = np.loadtxt(path, ndmin=2)
    if a.size == 0:
        return np.zeros(0), np.zeros((0, 3)), np.zeros((0, 4))
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def throttle(stamps, min_dt=0.1):
    """Indices the reference's recorder would keep (vo_repub_rec.cpp:77-78): `last_time` is a function-static set at the first
    call and never updated, so the poses of the first min_dt after the first call are dropped and EVERY later pose is written
    (it is a start-up delay, not a 10 Hz decimation)."""
    stamps = np.asarray(stamps, float)
    if len(stamps) == 0:
        return np.zeros(0, dtype=int)
    return np.nonzero(stamps - stamps[0] > min_dt)[0].astype(int)


# ---------------------------------------------------------------------------------------------- KITTI
def write_kitti(path, rotations, positions):
    with open(path, "w") as f:
        for R, t in zip(rotations, positions):
            f.write(" ".join("%.6g" % v for v in (R[0, 0], R[0, 1], R[0, 2], t[0], R[1, 0], R[1, 1], R[1, 2], t[1],
                                                  R[2, 0], R[2, 1], R[2, 2], t[2])) + "\n")


def _kitti_rows(text):
    a = np.loadtxt(io.StringIO(text), ndmin=2)
    assert a.shape[1] == 12, "KITTI pose files have 12 columns"
    M = a.reshape(-1, 3, 4)
    return M[:, :, :3].copy(), M[:, :, 3].copy()


def read_kitti(path, member=None):
    """-> (rotations [n,3,3], positions [n,3]).  `path` may be a poses.zip (give the member, e.g. 'poses/00.txt')."""
    if str(path).endswith(".zip"):
        with zipfile.ZipFile(path) as z:
            names = [n for n in z.namelist() if n.endswith(".txt")]
            name = member if member is not None else sorted(names)[0]
            return _kitti_rows(z.read(name).decode())
    return _kitti_rows(open(path).read())


# ---------------------------------------------------------------------------------------------- EuRoC ASL
def read_euroc_groundtruth(csv_path):
    """state_groundtruth_estimate0/data.csv -> (stamps [s], positions [n,3], quats_wxyz [n,4])."""
    rows = []
    for line in open(csv_path):
        if line.startswith("#") or not line.strip():
            continue
        v = line.split(",")
        rows.append([float(x) for x in v[:8]])
    a = np.array(rows).reshape(-1, 8)
    return a[:, 0] * 1e-9, a[:, 1:4], a[:, 4:8]


def read_euroc_image_list_ns(csv_path):
    """mav0/cam*/data.csv -> [(stamp_ns (int: 19 digits do not fit a double), filename)]."""
    out = []
    for line in open(csv_path):
        if line.startswith("#") or not line.strip():
            continue
        ts, name = line.strip().split(",")[:2]
        out.append((int(ts), name.strip()))
    return out


def read_euroc_image_list(csv_path):
    """mav0/cam*/data.csv -> [(stamp_s, filename)]."""
    return [(ns * 1e-9, name) for ns, name in read_euroc_image_list_ns(csv_path)]


def read_euroc_imu(csv_path):
    """mav0/imu0/data.csv -> [n,7] (t, gyro xyz, acc xyz) in the sensor frame (the reference remaps axes per type_of_vi)."""
    rows = []
    for line in open(csv_path):
        if line.startswith("#") or not line.strip():
            continue
        v = [float(x) for x in line.split(",")[:7]]
        rows.append([v[0] * 1e-9] + v[1:])
    return np.array(rows).reshape(-1, 7)


# ---------------------------------------------------------------------------------------------- ATE
def associate(stamps_a, stamps_b, max_dt=0.02):
    """Nearest-timestamp association (each stamp of b used at most once) -> (idx_a, idx_b)."""
    stamps_a, stamps_b = np.asarray(stamps_a, float), np.asarray(stamps_b, float)
    if len(stamps_a) == 0 or len(stamps_b) == 0:
        return np.zeros(0, int), np.zeros(0, int)
    order = np.argsort(stamps_b)
    sb = stamps_b[order]
    pos = np.searchsorted(sb, stamps_a)
    ia, ib, used = [], [], set()
    for i, p in enumerate(pos):
        best, bd = -1, max_dt
        for c in (p - 1, p):
            if 0 <= c < len(sb):
                d = abs(sb[c] - stamps_a[i])
                if d <= bd and order[c] not in used:
                    best, bd = order[c], d
        if best >= 0:
            used.add(best)
            ia.append(i)
            ib.append(best)
    return np.array(ia, int), np.array(ib, int)


def umeyama(src, dst, with_scale=False):
    """Least-squares similarity/rigid transform dst ~ s R src + t (Umeyama 1991) -> (s, R, t)."""
    src, dst = np.asarray(src, float), np.asarray(dst, float)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / (xs ** 2).sum() * len(src)) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ate_rmse(est_pos, ref_pos, align=True, with_scale=False):
    """RMSE of ||ref - (s R est + t)|| over associated positions."""
    est_pos, ref_pos = np.asarray(est_pos, float), np.asarray(ref_pos, float)
    if len(est_pos) < 3 and align:
        raise ValueError("need at least 3 poses to align")
    if align:
        s, R, t = umeyama(est_pos, ref_pos, with_scale)
        est_pos = (s * (R @ est_pos.T)).T + t
    return float(np.sqrt(np.mean(np.sum((est_pos - ref_pos) ** 2, axis=1))))


def ate_from_files(est_path, ref_path, max_dt=0.02, with_scale=False):
    te, pe, _ = read_stamped(est_path)
    tr, pr, _ = read_stamped(ref_path)
    ia, ib = associate(te, tr, max_dt)
    return ate_rmse(pe[ia], pr[ib], True, with_scale), len(ia)


# ---------------------------------------------------------------------------------------------- EuRoC ASL sequences
def sensor_to_flvis_imu(imu_type, acc, gyro):
    """The axis remap of TrackingNodeletClass::imu_callback (src/frontend/vo_tracking.cpp:331-357): sensor frame -> the
    FLVIS IMU frame.  imu_type 0 D435I, 1 EuRoC_MAV, 2 PIXHAWK.  (flvis_imu_feed applies the same remap inside the library;
    this copy is for feeding the CPU restatement from a dataset.)"""
    a, g = np.asarray(acc, float), np.asarray(gyro, float)
    if imu_type == 0:
        return np.array([-a[2], a[0], a[1]]), np.array([g[2], -g[0], -g[1]])
    if imu_type == 1:
        return np.array([-a[2], a[1], -a[0]]), np.array([g[2], -g[1], g[0]])
    return np.array([-a[0], -a[1], -a[2]]), np.array([g[0], g[1], g[2]])


def load_gray(path):
    """8-bit grayscale image file -> uint8 [h,w] (PIL; EuRoC ships 8-bit PNGs)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8).copy()


class EurocSequence:
    """An EuRoC ASL sequence folder (`<seq>/mav0/{cam0,cam1,imu0,state_groundtruth_estimate0}`): stereo pairs with equal
    stamps (what the reference's ExactTime synchroniser delivers, vo_tracking.cpp:308-319), the IMU samples between
    consecutive pairs, the ground truth if present."""

    def __init__(self, root):
        import os
        self.mav = os.path.join(root, "mav0") if os.path.isdir(os.path.join(root, "mav0")) else root
        c0 = dict(read_euroc_image_list_ns(os.path.join(self.mav, "cam0", "data.csv")))
        c1 = dict(read_euroc_image_list_ns(os.path.join(self.mav, "cam1", "data.csv")))
        self.stamps_ns = sorted(set(c0) & set(c1))
        self.files = [(os.path.join(self.mav, "cam0", "data", c0[k]), os.path.join(self.mav, "cam1", "data", c1[k]))
                      for k in self.stamps_ns]
        self.imu = read_euroc_imu(os.path.join(self.mav, "imu0", "data.csv"))     # [n,7]: t, gyro xyz, acc xyz (sensor frame)
        gt = os.path.join(self.mav, "state_groundtruth_estimate0", "data.csv")
        self.groundtruth = read_euroc_groundtruth(gt) if os.path.exists(gt) else None

    def __len__(self):
        return len(self.stamps_ns)

    def frames(self, first=0, count=None):
        """yields (t_seconds, img0, img1, imu_rows) where imu_rows [k,7] = (t, gyro xyz, acc xyz) with t_prev < t <= t_frame."""
        last = len(self) if count is None else min(len(self), first + count)
        ti = self.imu[:, 0] if len(self.imu) else np.zeros(0)
        t_prev = -np.inf if first == 0 else self.stamps_ns[first - 1] * 1e-9
        for k in range(first, last):
            t = self.stamps_ns[k] * 1e-9
            sel = (ti > t_prev) & (ti <= t)
            yield t, load_gray(self.files[k][0]), load_gray(self.files[k][1]), self.imu[sel]
            t_prev = t


class KittiSequence:
    """A KITTI odometry sequence folder as the reference's kitti_publisher reads it
    (src/independ_modules/kitti_publisher.cpp:100-131): `image_0/%06d.png` + `image_1/%06d.png` (gray stereo pair with the
    same index), published at a fixed rate (10 Hz there; `times.txt`, one stamp per line, is used when present), and an
    optional ground-truth file of 12-column poses T_w_c (first camera frame = world)."""

    def __init__(self, root, poses_file=None, rate_hz=10.0):
        import os
        self.root = root
        d0, d1 = os.path.join(root, "image_0"), os.path.join(root, "image_1")
        if not (os.path.isdir(d0) and os.path.isdir(d1)):
            raise FileNotFoundError("not a KITTI odometry sequence (image_0/ and image_1/ expected): %s" % root)
        n = 0
        while os.path.exists(os.path.join(d0, "%06d.png" % n)) and os.path.exists(os.path.join(d1, "%06d.png" % n)):
            n += 1
        self.files = [(os.path.join(d0, "%06d.png" % k), os.path.join(d1, "%06d.png" % k)) for k in range(n)]
        tp = os.path.join(root, "times.txt")
        if os.path.exists(tp):
            self.stamps = np.loadtxt(tp, ndmin=1)[:n]
        else:
            self.stamps = np.arange(n) / float(rate_hz)
        self.groundtruth = None
        if poses_file is None and os.path.exists(os.path.join(root, "poses.txt")):
            poses_file = os.path.join(root, "poses.txt")
        if poses_file is not None:
            R, t = read_kitti(poses_file)
            m = min(len(t), n)
            self.groundtruth = (self.stamps[:m], t[:m], R[:m])   # camera poses T_w_c

    def __len__(self):
        return len(self.files)

    def frames(self, first=0, count=None):
        """yields (t_seconds, img0, img1, imu_rows) -- imu_rows is always empty: the KITTI rig of the reference has no IMU."""
        last = len(self) if count is None else min(len(self), first + count)
        for k in range(first, last):
            yield float(self.stamps[k]), load_gray(self.files[k][0]), load_gray(self.files[k][1]), np.zeros((0, 7))


def open_sequence(root):
    """EuRoC ASL folder, KITTI odometry folder or a rosbag (.bag) with the reference's topics, by what it is."""
    import os
    if os.path.isfile(root) and root.endswith(".bag"):
        from .rosbag_io import RosbagSequence
        return RosbagSequence(root)
    if os.path.isdir(os.path.join(root, "image_0")):
        return KittiSequence(root)
    return EurocSequence(root)


def camera_to_body(positions_w_c, quats_wxyz_w_c, T_imu_cam44):
    """T_w_i = T_w_c * T_c_i for every pose (EuRoC ground truth is the body frame, the tracker reports the camera)."""
    T_c_i = np.linalg.inv(np.asarray(T_imu_cam44, float).reshape(4, 4))
    pos, quat = [], []
    for p, q in zip(positions_w_c, quats_wxyz_w_c):
        R = quat_to_rot(*q)
        pos.append(R @ T_c_i[:3, 3] + p)
        quat.append(rot_to_quat(R @ T_c_i[:3, :3]))
    return np.array(pos), np.array(quat)
