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


def epnp_numpy(Pw, z, K, signs=(1, 1, 1)):
    """EPnP (Lepetit, Moreno-Noguer, Fua 2009) as published, with LAPACK decompositions -- independent of csrc/epnp_core.hpp, which the
    device and the CPU checker share; `signs`: orientation of the three principal axes that carry the control points
    (an eigenvector's sign is the decomposition's choice)"""
    n = len(Pw)
    fu, fv, uc, vc = K
    z = ((z - K[2:]) / K[:2]).astype(np.float32).astype(np.float64) * K[:2] + K[2:]   # undistortPoints -> float, then x * fu + uc
    c0 = Pw.mean(0)
    d = Pw - c0
    ev, U = np.linalg.eigh(d.T @ d)
    cws = np.vstack([c0] + [c0 + sg * np.sqrt(max(ev[i], 0) / n) * U[:, i] for sg, i in zip(signs, (2, 1, 0))])
    al = np.linalg.solve((cws[1:] - cws[0]).T, (Pw - cws[0]).T).T
    al = np.hstack([1 - al.sum(1, keepdims=True), al])
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - z[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - z[:, 1])
    w, E = np.linalg.eigh(M.T @ M)
    v = E[:, :4].T.reshape(4, 4, 3)                     # v[i][control point]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[v[i, a] - v[i, b] for a, b in pairs] for i in range(4)])
    L = np.zeros((6, 10))
    for r in range(6):
        D = dv[:, r] @ dv[:, r].T
        L[r] = [D[0, 0], 2 * D[0, 1], D[1, 1], 2 * D[0, 2], 2 * D[1, 2], D[2, 2], 2 * D[0, 3], 2 * D[1, 3], 2 * D[2, 3], D[3, 3]]
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for a, b in pairs])

    def gauss_newton(b):
        b = b.copy()
        for _ in range(5):
            J = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                          L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                          L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                          L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], 1)
            bb = np.array([b[0] * b[0], b[0] * b[1], b[1] * b[1], b[0] * b[2], b[1] * b[2], b[2] * b[2], b[0] * b[3], b[1] * b[3], b[2] * b[3], b[3] * b[3]])
            Q_, R_ = np.linalg.qr(J)
            b += np.linalg.solve(R_, Q_.T @ (rho - L @ bb))
        return b

    def pose(b):
        ccs = np.einsum("i,ijk->jk", b, v)
        pcs = al @ ccs
        if pcs[0, 2] < 0:
            pcs = -pcs
        pc0, pw0 = pcs.mean(0), Pw.mean(0)
        Uu, _, Vt = np.linalg.svd((pcs - pc0).T @ (Pw - pw0))
        R = Uu @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        X = Pw @ R.T + t
        e = np.hypot(uc + fu * X[:, 0] / X[:, 2] - z[:, 0], vc + fv * X[:, 1] / X[:, 2] - z[:, 1]).mean()
        return e, R, t

    cands = []
    x = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]                      # N = 4
    b = np.array([np.sqrt(abs(x[0])), 0, 0, 0])
    b[1:] = np.sign(x[0] if x[0] != 0 else 1) * x[1:] / b[0]
    cands.append(pose(gauss_newton(b)))
    for cols in ([0, 1, 2], [0, 1, 2, 3, 4]):                                        # N = 2, N = 3
        x = np.linalg.lstsq(L[:, cols], rho, rcond=None)[0]
        if x[0] < 0:
            b = np.array([np.sqrt(-x[0]), np.sqrt(-x[2]) if x[2] < 0 else 0.0, 0, 0])
        else:
            b = np.array([np.sqrt(x[0]), np.sqrt(x[2]) if x[2] > 0 else 0.0, 0, 0])
        if x[1] < 0:
            b[0] = -b[0]
        if len(cols) == 5:
            b[2] = x[3] / b[0]
        cands.append(pose(gauss_newton(b)))
    best = cands[0]
    if cands[1][0] < best[0]:
        best = cands[1]
    if cands[2][0] < best[0]:
        best = cands[2]
    return best[1], best[2]


def gnu_sort(v, less):
    """std::sort as GNU libstdc++ implements it (introsort: 2 floor(log2 n) quicksort levels with a median-of-three pivot and an
    unguarded Hoare partition on ranges longer than 16, heap sort beyond that budget, one final insertion sort), in place on the
    Python list v.  Not stable: this is where the reference's `sort(..., sortbysecdesc)` (feature_dem.cpp:170,230) leaves candidates
    whose scores tie."""
    n = len(v)
    if n == 0:
        return v

    def adjust_heap(first, hole, length, value):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if less(v[first + child], v[first + child - 1]):
                child -= 1
            v[first + hole] = v[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[first + hole] = v[first + child - 1]
            hole = child - 1
        parent = int((hole - 1) / 2)
        while hole > top and less(v[first + parent], value):
            v[first + hole] = v[first + parent]
            hole = parent
            parent = int((hole - 1) / 2)
        v[first + hole] = value

    def heap_sort(first, last):
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, v[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            value = v[last]
            v[last] = v[first]
            adjust_heap(first, 0, last - first, value)

    def linear_insert(last):
        val = v[last]
        nxt = last - 1
        while less(val, v[nxt]):
            v[last] = v[nxt]
            last = nxt
            nxt -= 1
        v[last] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if less(v[i], v[first]):
                val = v[i]
                v[first + 1:i + 1] = v[first:i]
                v[first] = val
            else:
                linear_insert(i)

    def loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            a, b, c = first + 1, first + (last - first) // 2, last - 1
            if less(v[a], v[b]):
                m = b if less(v[b], v[c]) else (c if less(v[a], v[c]) else a)
            elif less(v[a], v[c]):
                m = a
            elif less(v[b], v[c]):
                m = c
            else:
                m = b
            v[first], v[m] = v[m], v[first]
            f, l = first + 1, last
            while True:
                while less(v[f], v[first]):
                    f += 1
                l -= 1
                while less(v[first], v[l]):
                    l -= 1
                if not f < l:
                    break
                v[f], v[l] = v[l], v[f]
                f += 1
            loop(f, last, depth)
            last = f

    loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            linear_insert(i)
    else:
        insertion_sort(0, n)
    return v

