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


def test_optimum_matches_an_independent_scipy_minimisation_of_the_same_robust_cost():
    """The same problem stated from scratch with rotation matrices (scipy Rotation) -- vertices T_w_c of keyframes kf_prev..kf_curr,
    EdgeSE3 error (translation, vector part of the w >= 0 quaternion) of X_i^-1 X_j against the measurement, Cauchy on each edge's
    squared error (delta 1, information I), first vertex fixed -- and minimised by scipy.optimize.least_squares on residuals whose
    squared norm per edge is log(1 + |e|^2).  The oracle's Levenberg (g2o's schedule) must land on the same optimum."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation

    p = PS.make_loop(11, n_kf=24, drift=(0.03, 0.006))
    n = len(p["est"])
    a, b = [int(x) for x in p["loops"][0]]
    r, T, drift, stats = pgo(p["est"], np.ones(n, np.uint8), p["loops"], p["loop_poses"])
    assert r == 1

    def to_Rt(p7):                                      # pose7 (t, q xyzw) -> (R, t)
        return Rotation.from_quat(p7[3:7]).as_matrix(), np.asarray(p7[:3], float)

    def inv(Rt):
        return Rt[0].T, -Rt[0].T @ Rt[1]

    def mul(A, B):
        return A[0] @ B[0], A[0] @ B[1] + A[1]

    ids = list(range(a, b + 1))                          # vertices; the first one (kf_prev) is fixed
    X0 = [inv(to_Rt(p["est"][i])) for i in ids]          # estimates T_w_c = T_c_w^-1
    edges = []
    for i in ids:
        for j in range(i + 1, min(b, i + 5) + 1):
            Tji = mul(to_Rt(p["est"][j]), inv(to_Rt(p["est"][i])))
            edges.append((i - a, j - a, inv(Tji)))       # measurement X_i^-1 X_j = (T_c_w,j T_w_c,i)^-1
    edges.append((0, b - a, inv(to_Rt(p["loop_poses"][0]))))

    def unpack(x):
        X = [X0[0]]
        for k in range(1, len(ids)):
            d = x[6 * (k - 1):6 * k]
            X.append(mul(X0[k], (Rotation.from_rotvec(d[3:]).as_matrix(), d[:3])))
        return X

    def edge_errors(X):
        out = []
        for i, j, Z in edges:
            E = mul(mul(inv(Z), inv(X[i])), X[j])
            q = Rotation.from_matrix(E[0]).as_quat()
            if q[3] < 0:
                q = -q
            out.append(np.concatenate([E[1], q[:3]]))
        return np.array(out)

    def residuals(x):
        e = edge_errors(unpack(x))
        s2 = (e * e).sum(1)
        scale = np.where(s2 > 1e-24, np.sqrt(np.log1p(s2) / np.maximum(s2, 1e-300)), 1.0)
        return (e * scale[:, None]).ravel()

    def cost_of(T_c_w):
        X = [inv(to_Rt(T_c_w[i])) for i in ids]
        e = edge_errors(X)
        return float(np.log1p((e * e).sum(1)).sum())

    sol = least_squares(residuals, np.zeros(6 * (len(ids) - 1)), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=200)
    cost_scipy = 2 * sol.cost
    cost_oracle = cost_of(T)
    assert abs(cost_oracle - stats[2]) < 1e-9 * max(1.0, stats[2])         # the oracle reports the cost this test defines
    assert cost_of(p["est"]) > 5 * cost_scipy                               # the drifted input is far from the optimum
    assert abs(cost_oracle - cost_scipy) < 1e-6 * cost_scipy + 1e-10, (cost_oracle, cost_scipy)
    Xs = unpack(sol.x)
    for k, i in enumerate(ids):
        Ro, to = inv(to_Rt(T[i]))
        assert np.linalg.norm(to - Xs[k][1]) < 1e-4 and np.linalg.norm(Ro - Xs[k][0]) < 1e-4, (i, np.linalg.norm(to - Xs[k][1]))
