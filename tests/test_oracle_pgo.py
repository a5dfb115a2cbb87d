"""CPU: the oracle's restatement of the reference's pose-graph optimisation (oracle/ref_pgo.cpp; vo_loopclosing.cpp:742-944, g2o
EdgeSE3 / VertexSE3 / Cauchy / Levenberg): edge Jacobians against central differences through the vertex update, and the
optimisation of a drifted loop."""
import ctypes as C

import numpy as np

import _geom as G
import _oracle as O
import _pgo_synth as PS


def _d(a):
    a = np.ascontiguousarray(a, np.float64)
    return a.ctypes.data_as(C.POINTER(C.c_double)), a


def edge(Xi, Xj, Z):
    e, Ji, Jj = np.zeros(6), np.zeros(36), np.zeros(36)
    a, _a = _d(Xi); b, _b = _d(Xj); z, _z = _d(Z)
    O.lib().ref_pgo_edge(a, b, z, e.ctypes.data_as(C.POINTER(C.c_double)), Ji.ctypes.data_as(C.POINTER(C.c_double)),
                         Jj.ctypes.data_as(C.POINTER(C.c_double)))
    return e, Ji.reshape(6, 6), Jj.reshape(6, 6)


def oplus(X, v):
    out = np.zeros(7)
    a, _a = _d(X); b, _b = _d(v)
    O.lib().ref_pgo_oplus(a, b, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def pgo(T_c_w, present, loops, loop_poses, iterations=100, initial_guess=True):
    T = np.ascontiguousarray(T_c_w, np.float64).copy()
    pres = np.ascontiguousarray(present, np.uint8)
    li = np.ascontiguousarray(loops, np.int32)
    lp = np.ascontiguousarray(loop_poses, np.float64)
    drift, stats = np.zeros(7), np.zeros(5)
    r = O.lib().ref_pgo_loop_closure(len(T), T.ctypes.data_as(C.POINTER(C.c_double)), pres.ctypes.data_as(C.POINTER(C.c_uint8)), len(li),
                                     li.ctypes.data_as(C.POINTER(C.c_int)), lp.ctypes.data_as(C.POINTER(C.c_double)), iterations,
                                     int(initial_guess), drift.ctypes.data_as(C.POINTER(C.c_double)),
                                     stats.ctypes.data_as(C.POINTER(C.c_double)))
    return r, T, drift, stats


def _rand_pose(rng, s=1.0):
    return G.pose7(G.rodrigues(rng.normal(0, 0.5, 3)), rng.normal(0, s, 3))


def test_edge_jacobians_match_central_differences():
    rng = np.random.default_rng(1)
    for trial in range(8):
        Xi, Xj = _rand_pose(rng, 2.0), _rand_pose(rng, 2.0)
        Z = PS.mul7(PS.mul7(PS.inv7(Xi), Xj), G.pose7(G.rodrigues(rng.normal(0, 0.2, 3)), rng.normal(0, 0.3, 3)))
        if trial == 7:      # an error rotation beyond 180 degrees: the w >= 0 normalisation flips the quaternion's sign
            Z = PS.mul7(PS.mul7(PS.inv7(Xi), Xj), G.pose7(G.rodrigues(np.array([0.0, 0.0, 3.4])), np.zeros(3)))
        e, Ji, Jj = edge(Xi, Xj, Z)
        h = 1e-6
        for k in range(6):
            v = np.zeros(6); v[k] = h
            ei_p, _, _ = edge(oplus(Xi, v), Xj, Z)
            ei_m, _, _ = edge(oplus(Xi, -v), Xj, Z)
            ej_p, _, _ = edge(Xi, oplus(Xj, v), Z)
            ej_m, _, _ = edge(Xi, oplus(Xj, -v), Z)
            assert np.allclose(Ji[:, k], (ei_p - ei_m) / (2 * h), atol=2e-7), (trial, k)
            assert np.allclose(Jj[:, k], (ej_p - ej_m) / (2 * h), atol=2e-7), (trial, k)
    # a measurement that agrees with the states: zero error
    e, _, _ = edge(Xi, Xj, PS.mul7(PS.inv7(Xi), Xj))
    assert np.abs(e).max() < 1e-14


def test_loop_closure_removes_the_drift():
    p = PS.make_loop(3)
    n = len(p["est"])
    present = np.ones(n, np.uint8)
    a, b = p["loops"][0]
    gap_before = PS.loop_gap(p["est"], p["gt"], a, b)
    r, T, drift, stats = pgo(p["est"], present, p["loops"], p["loop_poses"])
    assert r == 1 and stats[3] == n - 2 and stats[4] == 5 * (n - 2) - 15 + 1 and stats[0] >= 3
    assert np.array_equal(T[:2], p["est"][:2])                      # keyframes before the loop's first end are not touched
    assert np.allclose(T[2], p["est"][2], atol=1e-12)               # kf_prev is the fixed vertex
    assert stats[2] < 0.02 * stats[1]                               # robust chi2 after vs right after the initial guess
    gap_after = PS.loop_gap(T, p["gt"], a, b)
    # the accumulated drift between the two ends of the loop (decimetres) comes down to the accuracy of the loop measurement
    assert gap_before[0] > 0.1 and gap_after[0] < 0.1 * gap_before[0] and gap_after[1] < 0.9 * gap_before[1], (gap_before, gap_after)
    # (rotations: 66 x 5 unit-information odometry edges against ONE unit-information loop edge -- the chain gives way by about a third)
    # ... spread over the chain: no relative pose between neighbours moves by more than a few centimetres
    for k in range(3, n):
        assert PS.loop_gap(T, p["est"], k - 1, k)[0] < 0.05
    # drift of the last keyframe: T_c_w(new) = T_c_w(old) * drift  (vo_loopclosing.cpp:899-910, 922-925)
    assert np.allclose(PS.mul7(p["est"][n - 1], drift), T[n - 1], atol=1e-9)
    # without computeInitialGuess LM still gets to (about) the same optimum
    r2, T2, _, stats2 = pgo(p["est"], present, p["loops"], p["loop_poses"], initial_guess=False)
    assert abs(stats2[2] - stats[2]) < 1e-3 * max(stats[2], 1e-6) + 1e-6
    assert np.abs(T2 - T).max() < 1e-3


def test_several_loops_absent_keyframes_and_degenerate_inputs():
    p = PS.make_loop(5, n_kf=90, extra_loops=2)
    n = len(p["est"])
    present = np.ones(n, np.uint8)
    present[[20, 21, 47]] = 0                                       # kf_map_lc[i] == nullptr
    r, T, drift, stats = pgo(p["est"], present, p["loops"], p["loop_poses"])
    assert r == 1 and stats[3] == n - 2 - 3
    assert np.array_equal(T[[20, 21, 47]], p["est"][[20, 21, 47]])
    for (a, b) in p["loops"]:
        assert PS.loop_gap(T, p["gt"], a, b)[0] < 0.5 * PS.loop_gap(p["est"], p["gt"], a, b)[0]
    assert pgo(p["est"], present, np.zeros((0, 2), np.int32), np.zeros((0, 7)))[0] == 0
    # deterministic
    r, T_again, _, _ = pgo(p["est"], present, p["loops"], p["loop_poses"])
    assert np.array_equal(T, T_again)
