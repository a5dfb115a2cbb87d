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


@pytest.mark.parametrize("noise", [0.2, 0.5])
def test_epnp_device_agrees_with_the_independent_lapack_writeup(ctx, noise):
    """The device against a write-up that shares NO code with it (tests/_geom.py: numpy + LAPACK eigh / qr / svd / lstsq, the published
    algorithm): the bit-for-bit test above proves the device compile of epnp_core.hpp, this one proves the algorithm on the device.
    8 .. 700 correspondences (the chunk sums of 16 / 32 / 64 per chunk), tolerance 1e-8 on R and t.  Mirroring a control point along
    its principal axis is an equally valid configuration (an eigenvector's sign is the decomposition's choice), so the device pose has
    to coincide with ONE of the eight write-up poses."""
    import itertools
    import torch
    rng = np.random.default_rng(23)
    sizes = [8, 12, 30, 33, 64, 65, 200, 200, 300, 511, 700]
    sets = _sets(rng, sizes, noise)
    cap = 704
    p3 = np.zeros((len(sets), cap, 3), np.float32)
    p2 = np.zeros((len(sets), cap, 2), np.float32)
    cnt = np.zeros(len(sets), np.int32)
    for k, (P, z) in enumerate(sets):
        p3[k, :len(P)], p2[k, :len(P)], cnt[k] = P, z, len(P)
    out = ctx.debug_epnp(torch.from_numpy(p3).cuda(), torch.from_numpy(p2).cuda(), torch.from_numpy(cnt).cuda(), K4).cpu().numpy()
    for k, (P, z) in enumerate(sets):
        o = out[k]
        assert o[12] == 1.0, k
        Rg, tg = o[:9].reshape(3, 3), o[9:12]
        assert abs(np.linalg.det(Rg) - 1) < 1e-9
        d = []
        for sg in itertools.product((1, -1), repeat=3):
            Rn, tn = G.epnp_numpy(P.astype(np.float64), z.astype(np.float64), K4, sg)
            d.append(max(np.abs(Rg - Rn).max(), np.abs(tg - tn).max()))
        assert min(d) < 1e-8, ("set %d (n = %d)" % (k, len(P)), min(d))


def test_epnp_entry_point_refuses_what_it_cannot_hold(ctx):
    import torch
    import flvis_amd
    cnt = torch.tensor([5], dtype=torch.int32, device="cuda")
    with pytest.raises(flvis_amd.FlvisError):      # more than 1024 correspondences per set: capacity error, no launch
        ctx.debug_epnp(torch.zeros((1, 2048, 3), dtype=torch.float32, device="cuda"), torch.zeros((1, 2048, 2), dtype=torch.float32, device="cuda"), cnt, K4)
    rng = np.random.default_rng(3)
    (P, z), = _sets(rng, [8], 0.0)
    p3 = torch.zeros((3, 16, 3), dtype=torch.float32, device="cuda")
    p2 = torch.zeros((3, 16, 2), dtype=torch.float32, device="cuda")
    p3[:, :8], p2[:, :8] = torch.from_numpy(P).cuda(), torch.from_numpy(z).cuda()
    out = ctx.debug_epnp(p3, p2, torch.tensor([3, 8, 99], dtype=torch.int32, device="cuda"), K4).cpu().numpy()
    assert not out[0].any()                        # fewer than four correspondences: no pose, ok = 0
    ok, R, t = O.solve_epnp(P, z, K4)
    assert ok and out[1, 12] == 1.0 and np.array_equal(out[1, :9].reshape(3, 3), R)
    assert np.isfinite(out[2]).all() or out[2, 12] == 0.0   # a count beyond the capacity is clamped to it (16 points, 8 of them zeros)
