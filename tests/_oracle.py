"""ctypes loader for the CPU oracle (oracle/libflvis_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if os.environ.get("FLVIS_ORACLE_LIB"):  # bench.py's timing build (oracle/_fast, -O3 -march=native); never set by the tests
        _LIB = C.CDLL(os.environ["FLVIS_ORACLE_LIB"])
        return _LIB
    so = os.path.join(ROOT, "oracle", "libflvis_ref.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".cpp", ".h", ".hpp"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    _LIB = C.CDLL(so)
    return _LIB


def use_sum_order(order):
    """Switches every later call of this module to another build of the checker: "product" (the default: libflvis_ref.so, whose pose-LM /
    EPnP sums are the product's chunk sums) or "g2o" (libflvis_ref_g2o.so, `make -C oracle REF_ORDER=g2o`: the reference's own edge-by-
    edge / point-by-point order).  Objects created under one build must not be used under the other."""
    global _LIB
    lib()                                             # (builds both libraries when stale)
    name = {"product": "libflvis_ref.so", "g2o": "libflvis_ref_g2o.so", "solvers_product": "libflvis_ref_prod.so",
            "tail_cv": "libflvis_ref_cvtail.so"}[order]
    _LIB = C.CDLL(os.path.join(ROOT, "oracle", name))     # ("solvers_product": `make -C oracle SOLVERS=product`, the minimal solvers of rounds 1-5)
    assert _LIB.ref_sum_order() == (1 if order == "g2o" else 0)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def equalize_hist(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().ref_equalize_hist(_p(img, C.c_uint8), _p(out, C.c_uint8), img.shape[1], img.shape[0])
    return out


def cvt_bgr_to_gray(img):
    """img uint8 [h,w,3|4] (BGR / BGRA) -> gray [h,w]."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    out = np.empty((h, w), np.uint8)
    lib().ref_cvt_bgr_to_gray(_p(img, C.c_uint8), c, _p(out, C.c_uint8), w, h)
    return out


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().ref_pyr_down(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8))
    return out


def lk(prev, nxt, prev_pts, next_pts, win=31, max_level=10, max_iter=30, eps=1e-3, use_initial=True, min_eig=1e-4):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    npts = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
    n = pp.shape[0]
    st = np.zeros(n, np.uint8)
    f = lib().ref_calc_optical_flow_pyr_lk
    f.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_float),
                  C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                  C.c_int, C.c_float]
    f(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), prev.shape[1], prev.shape[0], _p(pp, C.c_float), _p(npts, C.c_float),
      _p(st, C.c_uint8), n, win, max_level, max_iter, eps, int(use_initial), min_eig)
    return npts, st


def min_eigen_map(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty(img.shape, np.float32)
    lib().ref_min_eigen_map(_p(img, C.c_uint8), img.shape[1], img.shape[0], _p(out, C.c_float))
    return out


def gftt(img, max_corners, q, min_dist):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((max(max_corners, 1), 2), np.float32)
    f = lib().ref_good_features_to_track
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_float)]
    n = f(_p(img, C.c_uint8), img.shape[1], img.shape[0], max_corners, q, min_dist, _p(out, C.c_float))
    return out[:n].copy()


def dem_detect(img, f_para, cap=4096):
    img = np.ascontiguousarray(img, np.uint8)
    fp = np.ascontiguousarray(f_para, np.float64)
    out = np.zeros((cap, 2), np.float32)
    n = lib().ref_feature_dem_detect(_p(img, C.c_uint8), img.shape[1], img.shape[0], _p(fp, C.c_double),
                                     _p(out, C.c_float), cap)
    return out[:n].copy()


def dem_redetect(img, f_para, existed, cap=4096):
    img = np.ascontiguousarray(img, np.uint8)
    fp = np.ascontiguousarray(f_para, np.float64)
    ex = np.ascontiguousarray(existed, np.float64).reshape(-1, 2)
    out = np.zeros((cap, 2), np.float32)
    n = lib().ref_feature_dem_redetect(_p(img, C.c_uint8), img.shape[1], img.shape[0], _p(fp, C.c_double),
                                       _p(ex, C.c_double), ex.shape[0], _p(out, C.c_float), cap)
    return out[:n].copy()


# ---------------------------------------------------------------------------------------------- geometry
def _d(a):
    return _p(np.ascontiguousarray(a, np.float64), C.c_double)


def poly_real_roots(coeffs):
    a = np.ascontiguousarray(coeffs, np.float64)
    r = np.zeros(4, np.float64)
    n = lib().ref_poly_real_roots(_p(a, C.c_double), len(a) - 1, _p(r, C.c_double))
    return r[:n].copy()


def project_points(p3d, pose7, K, D):
    p = np.ascontiguousarray(p3d, np.float32).reshape(-1, 3)
    out = np.zeros((len(p), 2), np.float32)
    lib().ref_project_points(_p(p, C.c_float), len(p), _d(pose7), _d(K), _d(D), _p(out, C.c_float))
    return out


def undistort_points(src, K, D, R, P):
    s = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    out = np.zeros_like(s)
    lib().ref_undistort_points(_p(s, C.c_float), len(s), _d(K), _d(D), _d(np.asarray(R).reshape(9)),
                               _d(np.asarray(P).reshape(12)), _p(out, C.c_float))
    return out


def triangulate_dlt(pt1, pt2, P1, P2):
    out = np.zeros(3)
    lib().ref_triangulate_dlt(_d(pt1), _d(pt2), _d(np.asarray(P1).reshape(12)), _d(np.asarray(P2).reshape(12)),
                              _p(out, C.c_double))
    return out


def seven_point(x1, x2):
    F = np.zeros((3, 9))
    n = lib().ref_seven_point(_d(x1), _d(x2), _p(F, C.c_double))
    return F[:n].reshape(n, 3, 3)


def find_fundamental_ransac(m1, m2, thr=5.0, conf=0.99, seed=1):
    a = np.ascontiguousarray(m1, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(m2, np.float32).reshape(-1, 2)
    mask = np.zeros(len(a), np.uint8)
    f = lib().ref_find_fundamental_ransac
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_double, C.c_double, C.c_uint64,
                  C.POINTER(C.c_uint8)]
    n = f(_p(a, C.c_float), _p(b, C.c_float), len(a), thr, conf, seed, _p(mask, C.c_uint8))
    return n, mask


def p3p(P, f):
    R = np.zeros((4, 3, 3))
    t = np.zeros((4, 3))
    n = lib().ref_p3p(_d(P), _d(f), _p(R, C.c_double), _p(t, C.c_double))
    return R[:n], t[:n]


def solve_epnp(p3d, p2d, K4):
    a = np.ascontiguousarray(p3d, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(p2d, np.float32).reshape(-1, 2)
    R, t = np.zeros((3, 3)), np.zeros(3)
    f = lib().ref_solve_epnp
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ok = f(_p(a, C.c_float), _p(b, C.c_float), len(a), _d(K4), _p(R, C.c_double), _p(t, C.c_double))
    return bool(ok), R, t


def epnp_last():
    """intermediate values of this thread's last solve_epnp: betas [3,4], err [3], v [4,12], L [6,10], rho [6]"""
    b, e, v, L, rho = np.zeros(12), np.zeros(3), np.zeros(48), np.zeros(60), np.zeros(6)
    f = lib().ref_epnp_last
    f.argtypes = [C.POINTER(C.c_double)] * 5
    f.restype = None
    f(_p(b, C.c_double), _p(e, C.c_double), _p(v, C.c_double), _p(L, C.c_double), _p(rho, C.c_double))
    return b.reshape(3, 4), e, v.reshape(4, 12), L.reshape(6, 10), rho


def epnp_jacobi12(A):
    A = np.ascontiguousarray(A, np.float64)
    ev, V = np.zeros(12), np.zeros((12, 12))
    f = lib().ref_epnp_jacobi12
    f.argtypes = [C.POINTER(C.c_double)] * 3
    f.restype = C.c_int
    sweeps = f(_p(A, C.c_double), _p(ev, C.c_double), _p(V, C.c_double))
    return ev, V, sweeps


def solve_pnp_ransac(p3d, p2d, K4, iterative, pose7=None, iterations=100, reproj=3.0, conf=0.99, seed=1):
    a = np.ascontiguousarray(p3d, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(p2d, np.float32).reshape(-1, 2)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0]) if pose7 is None else np.ascontiguousarray(pose7, np.float64).copy()
    mask = np.zeros(len(a), np.uint8)
    f = lib().ref_solve_pnp_ransac
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int,
                  C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint8)]
    n = f(_p(a, C.c_float), _p(b, C.c_float), len(a), _d(K4), int(iterative), iterations, reproj, conf, seed,
          _p(pose, C.c_double), _p(mask, C.c_uint8))
    return n, pose, mask


def optimize_in_frame(pose7, lm3d, lm2d, ids, K4):
    pose = np.ascontiguousarray(pose7, np.float64).copy()
    a = np.ascontiguousarray(lm3d, np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(lm2d, np.float64).reshape(-1, 2)
    i = np.ascontiguousarray(ids, np.int64)
    ok = lib().ref_optimize_in_frame(_p(pose, C.c_double), _p(a, C.c_double), _p(b, C.c_double), _p(i, C.c_int64),
                                     len(a), _d(K4))
    return bool(ok), pose


class LocalMap:
    def __init__(self, window, K4):
        self.window = window
        lib().ref_localmap_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().ref_localmap_create(window, _d(K4)))

    def __del__(self):
        try:
            if self.h:
                lib().ref_localmap_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def push(self, frame_id, pose7, lm_id, lm_2d, lm_3d, cap=8192):
        ids = np.ascontiguousarray(lm_id, np.int64)
        n = len(ids)
        ofid = C.c_int64(0)
        opose = np.zeros(7)
        ocnt = C.c_int(0)
        oid = np.zeros(cap, np.int64)
        o3d = np.zeros((cap, 3))
        oocnt = C.c_int(0)
        ooid = np.zeros(cap, np.int64)
        r = lib().ref_localmap_push(self.h, C.c_int64(frame_id), _d(pose7), n, _p(ids, C.c_int64), _d(lm_2d), _d(lm_3d),
                                    C.byref(ofid), _p(opose, C.c_double), C.byref(ocnt), _p(oid, C.c_int64),
                                    _p(o3d, C.c_double), cap, C.byref(oocnt), _p(ooid, C.c_int64), cap)
        if not r:
            return None
        return dict(frame_id=ofid.value, pose7=opose, lm_id=oid[:ocnt.value].copy(), lm_3d=o3d[:ocnt.value].copy(),
                    outlier_id=ooid[:oocnt.value].copy())

    def set_imu_factor(self, on, sigma_g, q_c_b_wxyz):
        q = np.ascontiguousarray(q_c_b_wxyz, np.float64)
        lib().ref_localmap_set_imu_factor(self.h, int(bool(on)), C.c_double(sigma_g), _p(q, C.c_double))

    def set_imu_factor_pos(self, sigma_a, t_c_b):
        t = np.ascontiguousarray(t_c_b, np.float64)
        lib().ref_localmap_set_imu_factor_pos(self.h, C.c_double(sigma_a), _p(t, C.c_double))

    def next_imu_pos(self, dp, va):
        """the position preintegration that comes with the NEXT pushed keyframe (after next_imu)"""
        a, b = np.ascontiguousarray(dp, np.float64), np.ascontiguousarray(va, np.float64)
        lib().ref_localmap_next_imu_pos(_p(a, C.c_double), _p(b, C.c_double))

    def next_imu(self, dq_wxyz, dt):
        """the gyro preintegration that comes with the NEXT pushed keyframe"""
        q = np.ascontiguousarray(dq_wxyz, np.float64)
        lib().ref_localmap_next_imu(_p(q, C.c_double), C.c_double(dt))

    def poses(self):
        p = np.zeros((self.window, 7))
        f = np.zeros(self.window, np.int32)
        pr = np.zeros(self.window, np.int32)
        lib().ref_localmap_poses(self.h, _p(p, C.c_double), _p(f, C.c_int32), _p(pr, C.c_int32))
        return p, f, pr


def ba_solve(poses7, fixed, lms3, e_pose, e_lm, e_uv, K4, it1=12, cull=True, it2=8):
    p = np.ascontiguousarray(poses7, np.float64).copy()
    l = np.ascontiguousarray(lms3, np.float64).copy()
    fx = np.ascontiguousarray(fixed, np.int32)
    ep = np.ascontiguousarray(e_pose, np.int32)
    el = np.ascontiguousarray(e_lm, np.int32)
    uv = np.ascontiguousarray(e_uv, np.float64)
    alive = np.zeros(len(ep), np.uint8)
    tr = np.zeros(3)
    f = lib().ref_ba_solve
    f.restype = C.c_double
    chi = f(len(p), _p(p, C.c_double), _p(fx, C.c_int32), len(l), _p(l, C.c_double), len(ep), _p(ep, C.c_int32),
            _p(el, C.c_int32), _p(uv, C.c_double), _d(K4), it1, int(cull), it2, _p(alive, C.c_uint8), _p(tr, C.c_double))
    return dict(poses7=p, lms3=l, alive=alive, chi2=chi, trace=tr)


# ---------------------------------------------------------------------------------------------- tracker / config
class RefConfig(C.Structure):
    _fields_ = [("type_of_vi", C.c_int), ("image_width", C.c_int), ("image_height", C.c_int),
                ("cam0_intrinsics", C.c_double * 4), ("cam0_distortion", C.c_double * 4),
                ("cam1_intrinsics", C.c_double * 4), ("cam1_distortion", C.c_double * 4),
                ("T_imu_cam0", C.c_double * 16), ("T_cam0_cam1", C.c_double * 16),
                ("vifusion_para", C.c_double * 6), ("feature_para", C.c_double * 6), ("dr_para", C.c_double * 3),
                ("window_size", C.c_int),
                ("cam_type", C.c_int), ("has_imu_type", C.c_int), ("skip_first_n_imgs", C.c_int),
                ("need_equal_hist", C.c_int),
                ("R0", C.c_double * 9), ("R1", C.c_double * 9), ("P0", C.c_double * 12), ("P1", C.c_double * 12),
                ("depth_factor", C.c_double)]


def load_config(path):
    cfg = RefConfig()
    assert lib().ref_config_sizeof() == C.sizeof(RefConfig), (lib().ref_config_sizeof(), C.sizeof(RefConfig))
    err = C.create_string_buffer(256)
    ok = lib().ref_config_load_yaml(path.encode(), C.byref(cfg), err, 256)
    if not ok:
        raise RuntimeError(err.value.decode())
    return cfg


class Tracker:
    def __init__(self, cfg, seed=0xF1715):
        lib().ref_tracker_create.restype = C.c_void_p
        lib().ref_tracker_create.argtypes = [C.POINTER(RefConfig), C.c_uint64]
        self.h = C.c_void_p(lib().ref_tracker_create(C.byref(cfg), seed))
        self.cfg = cfg

    def __del__(self):
        try:
            if self.h:
                lib().ref_tracker_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def imu(self, t, acc, gyro):
        out = np.zeros(10)
        lib().ref_tracker_imu(self.h, C.c_double(t), _d(acc), _d(gyro), _p(out, C.c_double))
        return out

    def image(self, t, img0, img1):
        img0 = np.ascontiguousarray(img0, np.uint8)
        # depth modes: the second image is the Z16 depth image (uint16), passed through the same pointer
        img1 = np.ascontiguousarray(img1) if img1.dtype == np.uint16 else np.ascontiguousarray(img1, np.uint8)
        st = C.c_int(0)
        pose = np.zeros(7)
        n = C.c_int(0)
        dbg = np.zeros(3, np.int32)
        fl = lib().ref_tracker_image(self.h, C.c_double(t), _p(img0, C.c_uint8), C.cast(img1.ctypes.data, C.POINTER(C.c_uint8)), C.byref(st),
                                     _p(pose, C.c_double), C.byref(n), _p(dbg, C.c_int32))
        return dict(new_keyframe=bool(fl & 1), reset_cmd=bool(fl & 2), state=st.value, pose7=pose, n_landmarks=n.value,
                    dbg=dbg)

    def stereo_depth(self, img0, img1, pt2d_plane, pt2d_undistort, pt3d_w, has_depth, pose7, rng):
        """CameraFrame::recover3DPts_c_FromStereo alone (camera_frame.cpp:93-180) on this tracker's rig and rand() generator:
        returns (pt3ds [n,3] float64, maskHas3DInf [n] uint8)."""
        n = len(pt2d_plane)
        p0 = np.ascontiguousarray(pt2d_plane, np.float32).reshape(-1, 2)
        p0u = np.ascontiguousarray(pt2d_undistort, np.float32).reshape(-1, 2)
        p3 = np.ascontiguousarray(pt3d_w, np.float32).reshape(-1, 3)
        has = np.ascontiguousarray(has_depth, np.uint8)
        T = np.ascontiguousarray(pose7, np.float64)
        out = np.zeros((n, 3))
        mask = np.zeros(n, np.uint8)
        img0, img1 = np.ascontiguousarray(img0), np.ascontiguousarray(img1)
        lib().ref_tracker_stereo_depth.restype = None
        lib().ref_tracker_stereo_depth.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p]
        lib().ref_tracker_stereo_depth(self.h, img0.ctypes.data, img1.ctypes.data, n, p0.ctypes.data, p0u.ctypes.data, p3.ctypes.data,
                                       has.ctypes.data, T.ctypes.data, C.c_float(rng), out.ctypes.data, mask.ctypes.data)
        return out, mask

    def landmarks(self, cap=2048):
        ids = np.zeros(cap, np.int64)
        p2d = np.zeros((cap, 2))
        p2u = np.zeros((cap, 2))
        p3w = np.zeros((cap, 3))
        fl = np.zeros(cap, np.uint8)
        n = lib().ref_tracker_landmarks(self.h, cap, _p(ids, C.c_int64), _p(p2d, C.c_double), _p(p2u, C.c_double),
                                        _p(p3w, C.c_double), _p(fl, C.c_uint8))
        return dict(ids=ids[:n].copy(), p2d=p2d[:n].copy(), p2u=p2u[:n].copy(), p3w=p3w[:n].copy(), flags=fl[:n].copy())

    def keyframe(self, cap=2048):
        fid = C.c_int64(0)
        pose = np.zeros(7)
        ids = np.zeros(cap, np.int64)
        p2u = np.zeros((cap, 2))
        p3w = np.zeros((cap, 3))
        n = lib().ref_tracker_keyframe(self.h, cap, C.byref(fid), _p(pose, C.c_double), _p(ids, C.c_int64),
                                       _p(p2u, C.c_double), _p(p3w, C.c_double))
        return dict(frame_id=fid.value, pose7=pose, lm_id=ids[:n].copy(), lm_2d=p2u[:n].copy(), lm_3d=p3w[:n].copy())

    def keyframe_imu(self):
        """(valid, dq (w, x, y, z), dt): the gyro preintegration attached to the last keyframe (an addition, see ref_tracking.hpp)"""
        dq = np.zeros(4)
        dt = C.c_double(0)
        v = lib().ref_tracker_keyframe_imu(self.h, _p(dq, C.c_double), C.byref(dt))
        return bool(v), dq, dt.value

    def keyframe_imu_pos(self):
        """(dp, va): the position preintegration attached to the last keyframe and the previous keyframe's filter velocity"""
        dp, va = np.zeros(3), np.zeros(3)
        lib().ref_tracker_keyframe_imu_pos(self.h, _p(dp, C.c_double), _p(va, C.c_double))
        return dp, va

    def correction_feed(self, frame_id, pose7, lm_id, lm_3d, outlier_id):
        """F2FTracking::correction_feed (dead in v2)."""
        lm_id = np.ascontiguousarray(lm_id, np.int64)
        lm_3d = np.ascontiguousarray(lm_3d, np.float64).reshape(-1, 3)
        outlier_id = np.ascontiguousarray(outlier_id, np.int64)
        pose7 = np.ascontiguousarray(pose7, np.float64)
        f = lib().ref_tracker_correction_feed
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                      C.c_int, C.POINTER(C.c_int64)]
        f(self.h, int(frame_id), _p(pose7, C.c_double), len(lm_id), _p(lm_id, C.c_int64), _p(lm_3d, C.c_double),
          len(outlier_id), _p(outlier_id, C.c_int64))

    def pose_records(self, cap=1024):
        rows = np.zeros((cap, 8))
        n = lib().ref_tracker_pose_records(self.h, cap, _p(rows, C.c_double))
        return rows[:n].copy()


# ------------------------------------------------------------------------------------------- ORB (oracle/ref_orb.cpp)
def resize_linear(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().ref_resize_linear_u8(_p(img, C.c_uint8), img.shape[1], img.shape[0], _p(out, C.c_uint8), int(dw), int(dh))
    return out


def fast_score_map(img, thr):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().ref_fast_score_map(_p(img, C.c_uint8), img.shape[1], img.shape[0], int(thr), _p(out, C.c_uint8))
    return out


def fast_detect(img, thr, cap=1 << 18):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((cap, 3), np.int32)
    n = lib().ref_fast_detect(_p(img, C.c_uint8), img.shape[1], img.shape[0], int(thr), _p(out, C.c_int), cap)
    assert n <= cap
    return out[:n].copy()


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().ref_gaussian_blur7(_p(img, C.c_uint8), img.shape[1], img.shape[0], _p(out, C.c_uint8))
    return out


def gauss_kernel7_fixed():
    k = np.zeros(7, np.int32)
    lib().ref_gauss_kernel7_fixed(_p(k, C.c_int))
    return k


def fast_atan2(y, x):
    f = lib().ref_fast_atan2
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_float]
    return f(y, x)


def orb_umax(half=15):
    u = np.zeros(half + 2, np.int32)
    lib().ref_orb_umax(half, _p(u, C.c_int))
    return u[:half + 1].copy()


def orb_ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    f = lib().ref_orb_ic_angle
    f.restype = C.c_float
    return f(_p(img, C.c_uint8), img.shape[1], int(x), int(y))


def orb_harris(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    f = lib().ref_orb_harris
    f.restype = C.c_float
    return f(_p(img, C.c_uint8), img.shape[1], int(x), int(y))


def orb_default_pattern():
    p = np.zeros((512, 2), np.int8)
    lib().ref_orb_default_pattern(_p(p, C.c_int8))
    return p


def orb_level_sizes(w, h, nlevels=8, sf=1.2):
    lw, lh, ls = np.zeros(nlevels, np.int32), np.zeros(nlevels, np.int32), np.zeros(nlevels, np.float32)
    f = lib().ref_orb_level_sizes
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    f(w, h, nlevels, sf, _p(lw, C.c_int), _p(lh, C.c_int), _p(ls, C.c_float))
    return lw, lh, ls


def orb_features_per_level(nfeatures=1000, nlevels=8, sf=1.2):
    n = np.zeros(nlevels, np.int32)
    f = lib().ref_orb_features_per_level
    f.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int)]
    f(nfeatures, nlevels, sf, _p(n, C.c_int))
    return n


def orb_detect_and_compute(img, nfeatures=1000, sf=1.2, nlevels=8, fast_thr=20, pattern=None, cap=8192, want_pyr=False):
    """-> kps [n,6] (x, y, size, angle, response, octave), desc [n,32] (, pyramid levels, blurred levels)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lw, lh, _ = orb_level_sizes(w, h, nlevels, sf)
    tot = int(np.sum(lw.astype(np.int64) * lh))
    kps = np.zeros((cap, 6), np.float32)
    desc = np.zeros((cap, 32), np.uint8)
    pyr = np.zeros(tot, np.uint8) if want_pyr else None
    blur = np.zeros(tot, np.uint8) if want_pyr else None
    pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8)
    f = lib().ref_orb_detect_and_compute
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p,
                  C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int, C.c_void_p, C.c_void_p]
    n = f(_p(img, C.c_uint8), w, h, nfeatures, sf, nlevels, fast_thr, None if pat is None else pat.ctypes.data,
          _p(kps, C.c_float), _p(desc, C.c_uint8), cap, None if pyr is None else pyr.ctypes.data,
          None if blur is None else blur.ctypes.data)
    assert n >= 0, "oracle ORB: capacity exceeded"
    if not want_pyr:
        return kps[:n].copy(), desc[:n].copy()
    lv, bl, off = [], [], 0
    for l in range(nlevels):
        sz = int(lw[l]) * int(lh[l])
        lv.append(pyr[off:off + sz].reshape(lh[l], lw[l]).copy())
        bl.append(blur[off:off + sz].reshape(lh[l], lw[l]).copy())
        off += sz
    return kps[:n].copy(), desc[:n].copy(), lv, bl


def hamming_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().ref_hamming_knn2(_p(q, C.c_uint8), len(q), _p(t, C.c_uint8), len(t), _p(idx, C.c_int), _p(dist, C.c_int))
    return idx, dist


def orb_match(a, b, ratio_max):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    pairs = np.zeros((max(len(a), 1), 2), np.int32)
    f = lib().ref_orb_match_mutual_ratio
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_double, C.POINTER(C.c_int)]
    n = f(_p(a, C.c_uint8), len(a), _p(b, C.c_uint8), len(b), float(ratio_max), _p(pairs, C.c_int))
    return pairs[:n].copy()


def lc_keyframe_landmarks(img0, img1, cam_type, kps, desc, P0=None, P1=None, K4=None):
    """ref_lc_keyframe_landmarks (vo_loopclosing.cpp:255-372): -> (lm_2d [k,2] float32, lm_3d [k,3], lm_desc [k,32])"""
    kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 6)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(kps)
    i1 = None if img1 is None else np.ascontiguousarray(img1, np.uint16 if cam_type == 2 else np.uint8)
    i0 = None if img0 is None else np.ascontiguousarray(img0, np.uint8)
    h, w = (i0 if i0 is not None else i1).shape
    dd = lambda a, m: None if a is None else np.ascontiguousarray(a, np.float64).reshape(m)
    p0, p1, k4 = dd(P0, 12), dd(P1, 12), dd(K4, 4)
    lm2 = np.zeros((max(n, 1), 2), np.float32)
    lm3 = np.zeros((max(n, 1), 3), np.float64)
    lmd = np.zeros((max(n, 1), 32), np.uint8)
    f = lib().ref_lc_keyframe_landmarks
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p]
    ptr = lambda a: None if a is None else a.ctypes.data
    k = f(ptr(i0), ptr(i1), w, h, int(cam_type), ptr(p0), ptr(p1), ptr(k4), ptr(kps), ptr(desc), n, ptr(lm2), ptr(lm3), ptr(lmd))
    return lm2[:k].copy(), lm3[:k].copy(), lmd[:k].copy()
