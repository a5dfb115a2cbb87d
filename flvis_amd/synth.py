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

    def T_c_w(self, t):
        """Ground-truth world->camera0 (R, t)."""
        R_w_c = self.R_w_i(t) @ R_I_C
        R_c_w = R_w_c.T
        return R_c_w, -R_c_w @ self.pos(t)


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
    def __init__(self, device="cpu", noise_sigma=2.0):
        self.dev = torch.device(device)
        self.tex = make_textures().to(self.dev)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
        self.rays_c = torch.stack([(xs - CX) / FX, (ys - CY) / FY, torch.ones_like(xs)], -1).to(self.dev)  # [H,W,3]
        self.noise_sigma = noise_sigma

    def render(self, R_w_c, c_w, seed=None):
        """R_w_c [S,3,3], c_w [S,3] (float64 tensors on device) -> uint8 [S,H,W]."""
        S = R_w_c.shape[0]
        d = torch.einsum("sij,hwj->shwi", R_w_c, self.rays_c)  # [S,H,W,3]
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
        return val.round().clamp(0, 255).to(torch.uint8)

    def stereo_frame(self, trajs, t, frame_idx=0):
        """Renders img0, img1 ([S,H,W] uint8 each) for all trajectories at time t."""
        Rs, c0, c1 = [], [], []
        for tr in trajs:
            R_w_c = tr.R_w_i(t) @ R_I_C
            p = tr.pos(t)
            Rs.append(R_w_c)
            c0.append(p)
            c1.append(p + R_w_c @ np.array([BASELINE, 0.0, 0.0]))
        Rw = torch.from_numpy(np.stack(Rs)).to(self.dev)
        i0 = self.render(Rw, torch.from_numpy(np.stack(c0)).to(self.dev), seed=2 * frame_idx)
        i1 = self.render(Rw, torch.from_numpy(np.stack(c1)).to(self.dev), seed=2 * frame_idx + 1)
        return i0, i1
