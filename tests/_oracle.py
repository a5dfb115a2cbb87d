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
    so = os.path.join(ROOT, "oracle", "libflvis_ref.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".cpp", ".h", ".hpp"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    _LIB = C.CDLL(so)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def equalize_hist(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().ref_equalize_hist(_p(img, C.c_uint8), _p(out, C.c_uint8), img.shape[1], img.shape[0])
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
