"""EPnP of the CPU restatement (oracle/ref_geom.cpp: solve_epnp, arithmetic in flvis_amd/csrc/epnp_core.hpp) against numpy: the 12 x 12 Jacobi
eigen-decomposition against numpy.linalg.eigh, the pose against the synthetic truth and against an independent numpy write-up of the
published algorithm (Lepetit et al. 2009, the steps cv::solvePnP(SOLVEPNP_EPNP) runs) that uses LAPACK for every decomposition."""
import itertools

import numpy as np
import pytest

import _geom as G
import _oracle as O

K4 = np.array([435.2, 435.2, 367.4, 252.2])


def test_jacobi12_matches_lapack():
    rng = np.random.default_rng(0)
    for rows in (10, 10, 12, 40, 400):          # 10 rows: the five-point kernel's M, two zero eigenvalues
        B = rng.normal(size=(rows, 12)) * rng.uniform(0.1, 30, 12)
        A = B.T @ B
        ev, V, sweeps = O.epnp_jacobi12(A)
        scale = np.trace(A) / 12
        # (the sweeps end after the first one whose rotations were all below 1e-7 of the matrix scale: what is left is ~ its square)
        assert sweeps <= 9
        assert np.abs(np.sort(ev) - np.linalg.eigvalsh(A)).max() < 1e-12 * scale
        assert np.abs(A @ V - V * ev).max() < 1e-9 * scale
        assert np.abs(V.T @ V - np.eye(12)).max() < 1e-14


def test_jacobi12_diagonal_and_repeated_eigenvalues():
    ev, V, sweeps = O.epnp_jacobi12(np.diag(np.arange(12.0)))
    assert sweeps == 1 and np.array_equal(V, np.eye(12)) and np.array_equal(ev, np.arange(12.0))
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.normal(size=(12, 12)))
    d = np.array([0, 0, 0, 1, 1, 2, 2, 2, 5, 5, 9, 9.0])
    A = (Q * d) @ Q.T
    A = (A + A.T) / 2
    ev, V, _ = O.epnp_jacobi12(A)
    assert np.abs(np.sort(ev) - d).max() < 1e-12 and np.abs(A @ V - V * ev).max() < 1e-8


def _scene(rng, n, noise):
    P, _ = G.random_scene(rng, n, K4)
    R = G.rodrigues(rng.normal(0, 0.2, 3))
    t = rng.normal(0, 0.3, 3)
    Pw = ((P - t) @ R).astype(np.float32).astype(np.float64)
    z = (G.project(R, t, Pw, K4) + rng.normal(0, noise, (n, 2))).astype(np.float32).astype(np.float64)
    return Pw, z, R, t


@pytest.mark.parametrize("n", [5, 6, 7, 12, 50, 200])
def test_epnp_recovers_the_pose_of_exact_data(n):
    rng = np.random.default_rng(n)
    good = 0
    for _ in range(20):
        Pw, z, R, t = _scene(rng, n, 0.0)
        ok, Rg, tg = O.solve_epnp(Pw, z, K4)
        assert ok and abs(np.linalg.det(Rg) - 1) < 1e-9
        # (pixel coordinates are float32: 3e-5 px; five points leave little redundancy)
        good += np.abs(Rg - R).max() < 2e-4 and np.abs(tg - t).max() < 2e-3
    assert good >= (17 if n < 7 else 20)


@pytest.mark.parametrize("n,noise", [(8, 0.2), (30, 0.5), (200, 0.5), (200, 2.0), (300, 0.5), (700, 0.5)])   # (chunk sums of 16 / 32 / 64)
def test_epnp_agrees_with_the_lapack_writeup_on_noisy_data(n, noise):
    rng = np.random.default_rng(100 + n)
    for _ in range(10 if n <= 200 else 3):
        Pw, z, R, t = _scene(rng, n, noise)
        ok, Rg, tg = O.solve_epnp(Pw, z, K4)
        assert ok and np.abs(Rg - R).max() < 0.05
        # the eigenvectors of MtM are well separated for n > 6 and the same steps follow: agreement far below the effect of the noise --
        # for the same control points.  Mirroring a control point along its axis is an equally valid configuration with a pose that
        # differs by a fraction of the noise's effect, so the restatement must coincide with ONE of the eight.
        d = []
        for sg in itertools.product((1, -1), repeat=3):
            Rn, tn = G.epnp_numpy(Pw, z, K4, sg)
            d.append(max(np.abs(Rg - Rn).max(), np.abs(tg - tn).max()))
        assert min(d) < 1e-8


def test_epnp_refuses_nothing_but_survives_degenerate_input():
    """all points in one place / on a line: cv::solvePnP returns some pose as well; the restatement must come back (no hang, no crash) and
    say so through ok = False or a finite pose"""
    Pw = np.tile(np.array([[0.3, -0.2, 4.0]]), (5, 1))
    z = np.tile(np.array([[300.0, 200.0]]), (5, 1))
    ok, R, t = O.solve_epnp(Pw, z, K4)
    assert (not ok) or (np.isfinite(R).all() and np.isfinite(t).all())
    Pw = np.array([[0, 0, 4 + 0.5 * i] for i in range(6)], float)
    z = G.project(np.eye(3), np.zeros(3), Pw, K4)
    ok, R, t = O.solve_epnp(Pw, z, K4)
    assert (not ok) or (np.isfinite(R).all() and np.isfinite(t).all())
