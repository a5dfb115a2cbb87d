"""Synthetic 640x480 stereo + 200 Hz IMU streams (harness code: data generation only, torch on CPU or GPU).

Scene: a textured box room (floor z=-1.5, ceiling z=+2.2, walls x=+4, y=+3, y=-3) rendered per camera by exact
ray/plane intersection with bilinear sampling of band-limited noise textures.  Camera model = the reference's
d435i-stereo yaml (launch/d435i/sn943222072828_stereo.yaml:12-25): fx=fy=384.16455078125, cx=320.2144470214844,
cy=238.94403076171875, zero distortion, T_cam0_cam1 = +0.05 m along x, T_imu_cam0 = [0 0 1; -1 0 0; 0 -1 0].
Trajectory per stream: Lissajous position, small yaw/roll/pitch oscillation; IMU = analytic derivatives
+ gravity 9.81 + white noise + constant bias.  World: x forward, y left, z up (the reference's init frame,
src/frontend/f2f_tracking.cpp:153-161).
"""
import math

import numpy as np
import torch

FX = FY = 384.16455078125
CX = 320.2144470214844
CY = 238.94403076171875
W, H = 640, 480
BASELINE = 0.05
FRAME_HZ = 20.0
IMU_HZ = 200.0
TEX = 1024
PPM = 160.0  # texture pixels per metre

R_I_C = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # T_imu_cam0 rotation (camera -> body)

D435I_STEREO_YAML = """type_of_vi: 3
image_width: 640
image_height: 480
cam0_intrinsics: [384.16455078125, 384.16455078125, 320.2144470214844, 238.94403076171875]
cam0_distortion_coeffs: [0.0, 0.0, 0.0, 0.0]
T_imu_cam0:
[ 0.0,  0.0,  1.0,  0.0,
 -1.0,  0.0,  0.0,  0.0,
  0.0, -1.0,  0.0,  0.0,
  0.0,  0.0,  0.0,  1.0]
cam1_intrinsics: [384.16455078125, 384.16455078125, 320.2144470214844, 238.94403076171875]
cam1_distortion_coeffs: [0.0, 0.0, 0.0, 0.0]
T_cam0_cam1:
[ 1.0,  0.0,  0.0,  0.05,
  0.0,  1.0,  0.0,  0.0,
  0.0,  0.0,  1.0,  0.0,
  0.0,  0.0,  0.0,  1.0]
is_lite_version:   True
vifusion_para1: 0.1
vifusion_para2: 0.03
vifusion_para3: 0.003
vifusion_para4: 0.01
vifusion_para5: 0.5
vifusion_para6: 0.1
feature_para1: 15
feature_para2: 30
feature_para3: 5
feature_para4: 500
feature_para5: 0.001
feature_para6: 5
dr_para1: 0.9
dr_para2: 20
dr_para3: 1.0
output_sparse_map: False
window_size:       8
"""

# depth-camera mode (type_of_vi 0, the key set of launch/d435i/sn943222072828_depth.yaml: cam0 only + depth_factor): the
# depth image is aligned to cam0, Z16 in millimetres
D435I_DEPTH_YAML = """type_of_vi: 0
image_width: 640
image_height: 480
cam0_intrinsics: [384.16455078125, 384.16455078125, 320.2144470214844, 238.94403076171875]
cam0_distortion_coeffs: [0.0, 0.0, 0.0, 0.0]
depth_factor: 1000.0
T_imu_cam0:
[ 0.0,  0.0,  1.0,  0.0,
 -1.0,  0.0,  0.0,  0.0,
  0.0, -1.0,  0.0,  0.0,
  0.0,  0.0,  0.0,  1.0]
is_lite_version:   True
vifusion_para1: 0.1
vifusion_para2: 0.01
vifusion_para3: 0.001
vifusion_para4: 0.001
vifusion_para5: 0.1
vifusion_para6: 0.1
feature_para1: 15
feature_para2: 30
feature_para3: 5
feature_para4: 500
feature_para5: 0.001
feature_para6: 5
dr_para1: 0.9
dr_para2: 8
dr_para3: 1.0
output_sparse_map: False
window_size:       8
"""


# EuRoC-MAV-like rig (type_of_vi 1: unrectified stereo with radial-tangential distortion, 752x480, equalizeHist): the
# public sensor calibration of the EuRoC VI sensor (cam0/cam1 sensor.yaml of the dataset) in FLVIS's parameter names.
_EUROC_T_IMU_MAVIMU = np.array([[0.0, 0.0, 1.0, 0.0], [0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
_EUROC_T_B_C0 = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                          [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                          [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                          [0.0, 0.0, 0.0, 1.0]])
_EUROC_T_B_C1 = np.array([[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
                          [0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024],
                          [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038],
                          [0.0, 0.0, 0.0, 1.0]])
_EUROC_K0 = (458.654, 457.296, 367.215, 248.375)
_EUROC_D0 = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
_EUROC_K1 = (457.587, 456.134, 379.999, 255.238)
_EUROC_D1 = (-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05)


def _fmt44(m):
    return "[" + ",\n ".join(", ".join("%.16g" % v for v in row) for row in m) + "]"


EUROC_LIKE_YAML = """type_of_vi: 1
image_width: 752
image_height: 480
T_imu_mavimu:
%s
cam0_intrinsics: [%s]
cam0_distortion_coeffs: [%s]
T_mavimu_cam0:
%s
cam1_intrinsics: [%s]
cam1_distortion_coeffs: [%s]
T_mavimu_cam1:
%s
is_lite_version:   False
vifusion_para1: 0.1
vifusion_para2: 0.01
vifusion_para3: 0.001
vifusion_para4: 0.001
vifusion_para5: 0.3
vifusion_para6: 0.1
feature_para1: 30
feature_para2: 20
feature_para3: 5
feature_para4: 1000
feature_para5: 0.01
feature_para6: 10
dr_para1: 0.90
dr_para2: 50
dr_para3: 1.0
output_sparse_map: True
window_size:       10
""" % (_fmt44(_EUROC_T_IMU_MAVIMU), ", ".join("%.16g" % v for v in _EUROC_K0), ", ".join("%.16g" % v for v in _EUROC_D0),
       _fmt44(_EUROC_T_B_C0), ", ".join("%.16g" % v for v in _EUROC_K1), ", ".join("%.16g" % v for v in _EUROC_D1),
       _fmt44(_EUROC_T_B_C1))


# KITTI-like rig (type_of_vi 4: rectified stereo given by two projection matrices, NO IMU; launch/KITTI/KITTI.yaml of the
# reference): KITTI's image size, intrinsics and feature parameters.  The baseline is 0.12 m instead of KITTI's 0.537 m because
# the synthetic room is 2-6 m deep (KITTI's street scenes are 5-50 m): disparities stay in KITTI's 15-45 px range.
KITTI_W, KITTI_H = 1241, 376
KITTI_FX, KITTI_CX, KITTI_CY = 718.856, 607.1928, 185.2157
KITTI_LIKE_BASELINE = 0.12
KITTI_LIKE_YAML = """type_of_vi: 4
image_width: 1241
image_height: 376
cam0_intrinsics: [718.856, 718.856, 607.1928, 185.2157]
cam0_distortion_coeffs: [0.0, 0.0, 0.0, 0.0]
cam1_intrinsics: [718.856, 718.856, 607.1928, 185.2157]
cam1_distortion_coeffs: [0.0, 0.0, 0.0, 0.0]
cam0_projection_matrix:
[718.856, 0.0,     607.1928, 0.0,
 0.0,     718.856, 185.2157, 0.0,
 0.0,     0.0,     1.0,      0.0,
 0.0,     0.0,     0.0,      0.0]
cam1_projection_matrix:
[718.856, 0.0,     607.1928, %.10g,
 0.0,     718.856, 185.2157, 0.0,
 0.0,     0.0,     1.0,      0.0,
 0.0,     0.0,     0.0,      0.0]
is_lite_version: False
vifusion_para1: 0.1
vifusion_para2: 0.03
vifusion_para3: 0.003
vifusion_para4: 0.01
vifusion_para5: 0.5
vifusion_para6: 0.1
feature_para1: 30
feature_para2: 15
feature_para3: 10
feature_para4: 2000
feature_para5: 0.0001
feature_para6: 10
dr_para1: 0.8
dr_para2: 1000
dr_para3: 0.0
output_sparse_map: False
window_size:       10
lcKFStart: 25
lcKFDist: 18
lcKFMaxDist: 50
lcKFLast: 20
lcNKFClosest: 2
ratioMax: 0.5
ratioRansac: 0.5
minPts: 20
minScore: 0.12
""" % (-KITTI_FX * KITTI_LIKE_BASELINE)


class Rig:
    """Stereo rig geometry for the renderer: intrinsics + radtan distortion per camera, camera0 -> body (R_i_c, t_i_c) and
    camera1 in camera0 coordinates (R_c0_c1, t_c0_c1)."""

    def __init__(self, width, height, K0, D0, K1, D1, T_i_c0, T_c0_c1):
        self.width, self.height = width, height
        self.K0, self.D0, self.K1, self.D1 = K0, D0, K1, D1
        self.R_i_c, self.t_i_c = T_i_c0[:3, :3].copy(), T_i_c0[:3, 3].copy()
        self.R_c0_c1, self.t_c0_c1 = T_c0_c1[:3, :3].copy(), T_c0_c1[:3, 3].copy()


def d435_rig():
    T_i_c = np.eye(4)
    T_i_c[:3, :3] = R_I_C
    T01 = np.eye(4)
    T01[0, 3] = BASELINE
    return Rig(W, H, (FX, FY, CX, CY), (0.0, 0.0, 0.0, 0.0), (FX, FY, CX, CY), (0.0, 0.0, 0.0, 0.0), T_i_c, T01)


def kitti_like_rig():
    """cam1 sits KITTI_LIKE_BASELINE to the right of cam0 (P1 = K [I | -b]); the camera looks along the body's x axis."""
    T_i_c = np.eye(4)
    T_i_c[:3, :3] = R_I_C
    T01 = np.eye(4)
    T01[0, 3] = KITTI_LIKE_BASELINE
    K = (KITTI_FX, KITTI_FX, KITTI_CX, KITTI_CY)
    return Rig(KITTI_W, KITTI_H, K, (0.0, 0.0, 0.0, 0.0), K, (0.0, 0.0, 0.0, 0.0), T_i_c, T01)


def euroc_rig():
    T_i_c0 = _EUROC_T_IMU_MAVIMU @ _EUROC_T_B_C0
    T01 = np.linalg.inv(_EUROC_T_B_C0) @ _EUROC_T_B_C1
    return Rig(752, 480, _EUROC_K0, _EUROC_D0, _EUROC_K1, _EUROC_D1, T_i_c0, T01)


def _pixel_rays(width, height, K, D):
    """Unit-depth rays of every pixel of a radtan camera: inverts the distortion by fixed-point iteration (as
    cv::undistortPoints does) so that the rendered image IS what the distorted camera would see."""
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float64), torch.arange(width, dtype=torch.float64), indexing="ij")
    xd, yd = (xs - K[2]) / K[0], (ys - K[3]) / K[1]
    k1, k2, p1, p2 = D
    x, y = xd.clone(), yd.clone()
    if any(abs(v) > 0 for v in D):
        for _ in range(40):
            r2 = x * x + y * y
            rad = 1 + k1 * r2 + k2 * r2 * r2
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (xd - dx) / rad, (yd - dy) / rad
    return torch.stack([x, y, torch.ones_like(x)], -1)


def make_textures(n_planes=5, seed=0xF1715000):
    """[n_planes, TEX, TEX] float32 in [0,255], periodic, band-limited (sum of octaves of bicubic value noise)."""
    out = []
    for p in range(n_planes):
        rng = np.random.default_rng(seed + p)
        acc = np.zeros((TEX, TEX), np.float64)
        amp, tot = 1.0, 0.0
        for cells in (8, 24, 64, 160):
            g = rng.random((cells, cells))
            g = np.concatenate([g, g[:, :3]], 1)
            g = np.concatenate([g, g[:3, :]], 0)  # periodic padding
            t = torch.from_numpy(g)[None, None]
            up = torch.nn.functional.interpolate(t, size=(int(TEX * (cells + 3) / cells), int(TEX * (cells + 3) / cells)),
                                                 mode="bicubic", align_corners=True)[0, 0, :TEX, :TEX].numpy()
            acc += amp * up
            tot += amp
            amp *= 0.65
        acc /= tot
        acc = (acc - acc.min()) / (acc.max() - acc.min())
        out.append((20 + 215 * acc).astype(np.float32))
    return torch.from_numpy(np.stack(out))


# planes: (normal n, offset d with n.x = d, tangent axes a, b)
_PLANES = [
    ((0.0, 0.0, 1.0), -1.5, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),   # floor
    ((0.0, 0.0, 1.0), 2.2, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),    # ceiling
    ((1.0, 0.0, 0.0), 4.0, (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)),    # front wall
    ((0.0, 1.0, 0.0), 3.0, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)),    # left wall
    ((0.0, 1.0, 0.0), -3.0, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)),   # right wall
]


def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


class Trajectory:
    """Body (IMU) pose T_w_i(t) of stream s, analytic derivatives for the IMU."""

    def __init__(self, s, n_streams_total=512, speed=1.0):
        self.phi = 2 * math.pi * s / n_streams_total
        self.k = speed

    def pos(self, t):
        k, ph = self.k, self.phi
        return np.array([1.5 * np.sin(0.4 * k * t), 1.0 * np.sin(0.3 * k * t + ph) - 1.0 * np.sin(ph), 0.3 * np.sin(0.5 * k * t)])

    def acc(self, t):
        k, ph = self.k, self.phi
        return np.array([-1.5 * (0.4 * k) ** 2 * np.sin(0.4 * k * t), -1.0 * (0.3 * k) ** 2 * np.sin(0.3 * k * t + ph),
                         -0.3 * (0.5 * k) ** 2 * np.sin(0.5 * k * t)])

    def ypr(self, t):
        k, ph = self.k, self.phi
        return (0.3 * np.sin(0.2 * k * t), 0.05 * np.sin(0.7 * k * t + ph) - 0.05 * np.sin(ph),
                0.05 * np.sin(0.7 * k * t + ph + 1.0) - 0.05 * np.sin(ph + 1.0))

    def ypr_dot(self, t):
        k, ph = self.k, self.phi
        return (0.3 * 0.2 * k * np.cos(0.2 * k * t), 0.05 * 0.7 * k * np.cos(0.7 * k * t + ph),
                0.05 * 0.7 * k * np.cos(0.7 * k * t + ph + 1.0))

    def R_w_i(self, t):
        y, p, r = self.ypr(t)
        return _rot_zyx(y, p, r)

    def omega_body(self, t):
        y, p, r = self.ypr(t)
        yd, pd, rd = self.ypr_dot(t)
        return np.array([rd - yd * np.sin(p), pd * np.cos(r) + yd * np.sin(r) * np.cos(p),
                         -pd * np.sin(r) + yd * np.cos(r) * np.cos(p)])

    def T_c_w(self, t, rig=None):
        """Ground-truth world->camera0 (R, t)."""
        R_i_c = R_I_C if rig is None else rig.R_i_c
        t_i_c = np.zeros(3) if rig is None else rig.t_i_c
        R_w_c = self.R_w_i(t) @ R_i_c
        c = self.pos(t) + self.R_w_i(t) @ t_i_c
        R_c_w = R_w_c.T
        return R_c_w, -R_c_w @ c


def imu_samples(traj, s, t0, t1, noise=True):
    """IMU samples with t0 < t <= t1 in the FLVIS IMU frame (what F2FTracking::imu_feed receives after the axis remap
    of src/frontend/vo_tracking.cpp:331-357): acc = R^T (a_w + (0,0,-9.81)), gyro = body rate.  Returns [n,7] (t, acc, gyro)."""
    k0 = int(math.floor(t0 * IMU_HZ + 1e-9)) + 1
    k1 = int(math.floor(t1 * IMU_HZ + 1e-9))
    out = []
    ba = np.array([0.05, -0.03, 0.02])
    bg = np.array([0.002, -0.001, 0.0015])
    for k in range(k0, k1 + 1):
        t = k / IMU_HZ
        R = traj.R_w_i(t)
        acc = R.T @ (traj.acc(t) + np.array([0.0, 0.0, -9.81]))
        gyro = traj.omega_body(t)
        if noise:
            rng = np.random.default_rng((0x1A2B0000 + s) * 100003 + k)
            acc = acc + ba + rng.normal(0, 0.02, 3)
            gyro = gyro + bg + rng.normal(0, 0.002, 3)
        out.append(np.concatenate([[t], acc, gyro]))
    return np.array(out).reshape(-1, 7)


def flvis_to_d435i_sensor(acc_f, gyro_f):
    """Inverse of the D435i axis remap (vo_tracking.cpp:333-340): returns sensor-frame (acc, gyro)."""
    acc_s = np.array([acc_f[1], acc_f[2], -acc_f[0]])
    gyro_s = np.array([-gyro_f[1], -gyro_f[2], gyro_f[0]])
    return acc_s, gyro_s


class Renderer:
    def __init__(self, device="cpu", noise_sigma=2.0, rig=None):
        self.dev = torch.device(device)
        self.tex = make_textures().to(self.dev)
        self.rig = rig if rig is not None else d435_rig()
        g = self.rig
        self.rays = [_pixel_rays(g.width, g.height, g.K0, g.D0).to(self.dev), _pixel_rays(g.width, g.height, g.K1, g.D1).to(self.dev)]
        self.rays_c = self.rays[0]  # [H,W,3]
        self.noise_sigma = noise_sigma

    def render(self, R_w_c, c_w, seed=None, cam=0, want_depth=False):
        """R_w_c [S,3,3], c_w [S,3] (float64 tensors on device) -> uint8 [S,H,W] as seen by camera `cam` of the rig
        (and, with want_depth, the z-depth [S,H,W] in metres: the rays have unit z, so the ray parameter IS the depth)."""
        S = R_w_c.shape[0]
        H, W = self.rig.height, self.rig.width
        d = torch.einsum("sij,hwj->shwi", R_w_c, self.rays[cam])  # [S,H,W,3]
        best = torch.full((S, H, W), 1e30, dtype=torch.float64, device=self.dev)
        val = torch.full((S, H, W), 128.0, dtype=torch.float32, device=self.dev)
        for pi, (n, off, a, b) in enumerate(_PLANES):
            n_t = torch.tensor(n, dtype=torch.float64, device=self.dev)
            a_t = torch.tensor(a, dtype=torch.float64, device=self.dev)
            b_t = torch.tensor(b, dtype=torch.float64, device=self.dev)
            dn = (d * n_t).sum(-1)
            lam = (off - (c_w * n_t).sum(-1))[:, None, None] / dn
            hit = (lam > 1e-6) & (lam < best) & torch.isfinite(lam)
            X = c_w[:, None, None, :] + lam[..., None] * d
            u = ((X * a_t).sum(-1) * PPM) % TEX
            v = ((X * b_t).sum(-1) * PPM) % TEX
            u0 = torch.floor(u)
            v0 = torch.floor(v)
            fu = (u - u0).to(torch.float32)
            fv = (v - v0).to(torch.float32)
            u0 = u0.long() % TEX
            v0 = v0.long() % TEX
            u1 = (u0 + 1) % TEX
            v1 = (v0 + 1) % TEX
            t = self.tex[pi]
            samp = (t[v0, u0] * (1 - fu) + t[v0, u1] * fu) * (1 - fv) + (t[v1, u0] * (1 - fu) + t[v1, u1] * fu) * fv
            val = torch.where(hit, samp, val)
            best = torch.where(hit, lam, best)
        if self.noise_sigma > 0:
            g = torch.Generator(device=self.dev)
            g.manual_seed(0x5EED0000 + (seed or 0))
            val = val + self.noise_sigma * torch.randn(val.shape, generator=g, device=self.dev, dtype=torch.float32)
        img = val.round().clamp(0, 255).to(torch.uint8)
        return (img, best) if want_depth else img

    def depth_frame(self, trajs, t, frame_idx=0, depth_factor=1000.0, max_range=None):
        """img0 uint8 [S,H,W] and the aligned Z16 depth image [S,H,W] (int16 storage, read as uint16) (depth * depth_factor, rounded; 0 beyond
        max_range, like a depth sensor's no-return value) -- the input pair of the depth-camera modes."""
        g = self.rig
        R0, c0 = [], []
        for tr in trajs:
            R_w_i = tr.R_w_i(t)
            R0.append(R_w_i @ g.R_i_c)
            c0.append(tr.pos(t) + R_w_i @ g.t_i_c)
        i0, z = self.render(torch.from_numpy(np.stack(R0)).to(self.dev), torch.from_numpy(np.stack(c0)).to(self.dev),
                            seed=2 * frame_idx, cam=0, want_depth=True)
        if max_range is not None:
            z = torch.where(z <= max_range, z, torch.zeros_like(z))
        # torch has few uint16 kernels: int16 storage with the same bit pattern (values < 32768, i.e. < 32 m at 1 mm units)
        d16 = (z * depth_factor).round().clamp(0, 32767).to(torch.int16)
        return i0, d16

    def stereo_frame(self, trajs, t, frame_idx=0):
        """Renders img0, img1 ([S,H,W] uint8 each) for all trajectories at time t."""
        g = self.rig
        R0, R1, c0, c1 = [], [], [], []
        for tr in trajs:
            R_w_i = tr.R_w_i(t)
            R_w_c0 = R_w_i @ g.R_i_c
            p0 = tr.pos(t) + R_w_i @ g.t_i_c
            R0.append(R_w_c0)
            c0.append(p0)
            R1.append(R_w_c0 @ g.R_c0_c1)
            c1.append(p0 + R_w_c0 @ g.t_c0_c1)
        i0 = self.render(torch.from_numpy(np.stack(R0)).to(self.dev), torch.from_numpy(np.stack(c0)).to(self.dev), seed=2 * frame_idx, cam=0)
        i1 = self.render(torch.from_numpy(np.stack(R1)).to(self.dev), torch.from_numpy(np.stack(c1)).to(self.dev), seed=2 * frame_idx + 1, cam=1)
        return i0, i1
