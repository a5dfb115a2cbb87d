"""flvis_amd/csrc/det_math.hpp: the deterministic sin / cos / atan / atan2 / log both the HIP kernels and the oracle execute
(same IEEE operations in the same order on both sides, so the closed-loop front-end can be compared bit for bit).  Here: their
accuracy against libm on the ranges the path uses (< 1 ulp; they are the fdlibm algorithms) and the special values."""
import ctypes as C
import math

import numpy as np

import _oracle as O


def _batch(which, a, b=None):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(a if b is None else b, np.float64)
    out = np.zeros_like(a)
    O.lib().ref_det_batch(which, len(a), a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                          out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def _ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_sin_cos_within_one_ulp_of_libm():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-7, 7, 200000), rng.uniform(-1e-3, 1e-3, 20000), rng.uniform(-400, 400, 50000),
                        np.linspace(-2 * math.pi, 2 * math.pi, 4001), [0.0, -0.0, 1e-300, math.pi / 4, -math.pi / 4, 1e5]])
    for which, f in ((0, np.sin), (1, np.cos)):
        got, want = _batch(which, x), f(x)
        nz = want != 0
        assert _ulps(got[nz], want[nz]).max() <= 1.0
        assert np.array_equal(got[~nz], want[~nz])


def test_atan_atan2_log_within_one_ulp_of_libm():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-5, 5, 100000), rng.uniform(-1e-4, 1e-4, 10000), 10.0 ** rng.uniform(-8, 8, 20000),
                        -(10.0 ** rng.uniform(-8, 8, 20000))])
    got, want = _batch(2, x), np.arctan(x)
    assert _ulps(got, want).max() <= 1.0
    y, xx = rng.normal(size=200000), rng.normal(size=200000)
    got, want = _batch(3, y, xx), np.arctan2(y, xx)
    assert _ulps(got, want).max() <= 1.0
    pos = np.concatenate([rng.uniform(1e-6, 10, 100000), 10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0.99, 1.01, 20000)])
    got, want = _batch(4, pos), np.log(pos)
    nz = want != 0
    assert _ulps(got[nz], want[nz]).max() <= 1.0


def test_acos_cbrt_within_one_ulp_of_libm():
    """det_acos / det_cbrt (round 6: the closed-form cubic and quartic of the OpenCV-shaped 7-point and P3P solvers)"""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, 200000), rng.uniform(-1e-9, 1e-9, 1000), 1 - 10.0 ** rng.uniform(-16, -1, 20000),
                        -1 + 10.0 ** rng.uniform(-16, -1, 20000), [0.0, 0.5, -0.5, 1.0, -1.0, 0.4999999, 0.5000001]])
    got, want = _batch(5, x), np.arccos(x)
    nz = want != 0
    assert _ulps(got[nz], want[nz]).max() <= 1.0
    assert np.array_equal(got[~nz], want[~nz])
    assert np.isnan(_batch(5, np.array([1.0000001, -2.0]))).all()
    y = np.concatenate([rng.uniform(-10, 10, 100000), 10.0 ** rng.uniform(-300, 300, 50000), -(10.0 ** rng.uniform(-300, 300, 50000)),
                        [8.0, -27.0, 1e-320, -1e-310]])
    got, want = _batch(6, y), np.cbrt(y)
    assert _ulps(got, want).max() <= 1.0
    assert np.array_equal(_batch(6, np.array([0.0, -0.0, math.inf, -math.inf])), np.array([0.0, -0.0, math.inf, -math.inf]))
    assert math.copysign(1, _batch(6, np.array([-0.0]))[0]) == -1.0


def test_special_values():
    L = O.lib()
    for f in (L.ref_det_sin, L.ref_det_cos, L.ref_det_atan, L.ref_det_log, L.ref_det_atan2, L.ref_det_powi):
        f.restype = C.c_double
    L.ref_det_atan2.argtypes = [C.c_double, C.c_double]
    L.ref_det_powi.argtypes = [C.c_double, C.c_int]
    for f in (L.ref_det_sin, L.ref_det_cos, L.ref_det_atan, L.ref_det_log):
        f.argtypes = [C.c_double]
    assert L.ref_det_sin(0.0) == 0.0 and L.ref_det_cos(0.0) == 1.0 and L.ref_det_atan(0.0) == 0.0 and L.ref_det_log(1.0) == 0.0
    assert math.copysign(1, L.ref_det_sin(-0.0)) == -1.0
    assert L.ref_det_log(0.0) == -math.inf and math.isnan(L.ref_det_log(-1.0)) and math.isnan(L.ref_det_sin(math.inf))
    assert L.ref_det_atan(math.inf) == math.atan(math.inf) and L.ref_det_atan(-math.inf) == math.atan(-math.inf)
    for y, x in ((0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (1.0, 1.0), (-3.0, -4.0), (2.0, -math.inf),
                 (math.inf, math.inf), (1e-310, 1.0)):
        assert L.ref_det_atan2(y, x) == math.atan2(y, x), (y, x)
    assert L.ref_det_powi(1.5, 3) == 1.5 * 1.5 * 1.5 and L.ref_det_powi(0.9, 7) == ((((((0.9 * 0.9) * 0.9) * 0.9) * 0.9) * 0.9) * 0.9)
    assert L.ref_det_powi(2.0, 0) == 1.0
