"""EPnP on the device (flvis_hip_debug_epnp: epnp_core.hpp run by one wavefront) against the CPU restatement (the same header run by
one lane): every intermediate value and the pose, bit for bit.  The restatement itself is pinned in test_oracle_epnp.py."""
import numpy as np
import pytest

import _geom as G
import _oracle as O

pytestmark = pytest.mark.gpu

K4 = np.array([435.2, 435.2, 367.4, 252.2])


@pytest.fixture(scope="module")
def ctx():
    import flvis_amd
    return flvis_amd.Context(0)


def _sets(rng, sizes, noise):
    out = []
    for n in sizes:
        P, _ = G.random_scene(rng, n, K4)
        R = G.rodrigues(rng.normal(0, 0.2, 3))
        t = rng.normal(0, 0.3, 3)
        Pw = ((P - t) @ R).astype(np.float32)
        z = (G.project(R, t, Pw.astype(np.float64), K4) + rng.normal(0, noise, (n, 2))).astype(np.float32)
        out.append((Pw, z))
    return out


@pytest.mark.parametrize("noise", [0.0, 0.5])
def test_epnp_device_equals_the_restatement_bit_for_bit(ctx, noise):
    import torch
    rng = np.random.default_rng(17)
    sizes = [5] * 24 + [4, 6, 7, 8, 12, 33, 64, 65, 200, 511, 700]
    sets = _sets(rng, sizes, noise)
    cap = 704
    p3 = np.zeros((len(sets), cap, 3), np.float32)
    p2 = np.zeros((len(sets), cap, 2), np.float32)
    cnt = np.zeros(len(sets), np.int32)
    for k, (P, z) in enumerate(sets):
        p3[k, :len(P)], p2[k, :len(P)], cnt[k] = P, z, len(P)
    out = ctx.debug_epnp(torch.from_numpy(p3).cuda(), torch.from_numpy(p2).cuda(), torch.from_numpy(cnt).cuda(), K4).cpu().numpy()
    for k, (P, z) in enumerate(sets):
        ok, R, t = O.solve_epnp(P, z, K4)
        betas, err, v, L, rho = O.epnp_last()
        o = out[k]
        what = "set %d (n = %d)" % (k, len(P))
        assert np.array_equal(o[136:142], rho), what
        assert np.array_equal(o[28:76].reshape(4, 12), v), (what, np.abs(o[28:76].reshape(4, 12) - v).max())
        assert np.array_equal(o[76:136].reshape(6, 10), L), what
        assert np.array_equal(o[13:25].reshape(3, 4), betas), (what, o[13:25].reshape(3, 4) - betas)
        assert np.array_equal(o[25:28], err), what
        assert bool(o[12]) == ok and np.array_equal(o[:9].reshape(3, 3), R) and np.array_equal(o[9:12], t), what
