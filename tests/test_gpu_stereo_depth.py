"""GPU parity of flvis_hip_stereo_depth -- CameraFrame::recover3DPts_c_FromStereo (camera_frame.cpp:93-180) in one call, the kernel-level
drop-in of BASELINE configs[1] -- against the oracle's restatement of the same function: masks, 3-D points and the
consumption of the rand() generator bit-exact, through the C ABI."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import _oracle as O
import _stereo_inputs as SI

pytestmark = pytest.mark.gpu


def _cfgs(text, tag):
    import flvis_amd
    p = os.path.join(tempfile.gettempdir(), "flvis_sd_%s.yaml" % tag)
    open(p, "w").write(text)
    cfg = flvis_amd.load_config(p)
    ocfg = O.RefConfig()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg))
    return cfg, ocfg


@pytest.mark.parametrize("rig_name", ["d435", "euroc"])
def test_stereo_depth_parity(rig_name):
    import flvis_amd
    import torch
    from flvis_amd import synth
    if rig_name == "d435":
        cfg, ocfg = _cfgs(synth.D435I_STEREO_YAML, "d435")
        rig, nframes, rngs = None, 50 + 6, (3.0, 2.0)          # the second call's short range fails the z test: dummy depths
    else:
        cfg, ocfg = _cfgs(synth.EUROC_LIKE_YAML, "euroc")      # unrectified pair: undistortPoints with D1, R1, P1 matters
        rig, nframes, rngs = synth.euroc_rig(), 6, (8.0, 8.0)
    sets = [SI.tracked_frame(ocfg, rig, s, nframes + k, device="cuda") for k, s in enumerate((3, 140))]
    n_sets, cap = 4, 640                                        # set 2 is empty, set 3 is set 0 cut to 17 landmarks (ragged counts)
    sets = sets + [None, {k: (v[:17] if k in ("p2d", "p2u", "p3w", "has") else v) for k, v in sets[0].items()}]
    h, w = sets[0]["img0"].shape
    img0 = np.zeros((n_sets, h, w), np.uint8)
    img1 = np.zeros((n_sets, h, w), np.uint8)
    p2d = np.zeros((n_sets, cap, 2), np.float32)
    p2u = np.zeros((n_sets, cap, 2), np.float32)
    p3w = np.zeros((n_sets, cap, 3), np.float32)
    has = np.zeros((n_sets, cap), np.uint8)
    cnt = np.zeros(n_sets, np.int32)
    poses = np.tile(np.array([0, 0, 0, 0, 0, 0, 1.0]), (n_sets, 1))
    for s, d in enumerate(sets):
        if d is None:
            continue
        n = len(d["p2d"])
        assert 17 <= n <= cap
        img0[s], img1[s], cnt[s], poses[s] = d["img0"], d["img1"], n, d["pose7"]
        p2d[s, :n], p2u[s, :n], p3w[s, :n], has[s, :n] = d["p2d"], d["p2u"], d["p3w"], d["has"]
    ctx = flvis_amd.Context(0)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    state = ctx.rand_seed(1, n_sets)
    refs = [O.Tracker(ocfg, 1) for _ in range(n_sets)]          # fresh rand() generators (seed 1, like the reference's process)
    seen_fail = seen_ok = 0
    for call in range(2):                                        # two calls: the generator state carries over on both sides
        rng = rngs[call]
        got3, gotm = ctx.stereo_depth(cfg, dev(img0), dev(img1), dev(p2d), dev(p2u), dev(p3w), dev(has), dev(cnt), poses, rng, state)
        got3, gotm = got3.cpu().numpy(), gotm.cpu().numpy()
        for s, d in enumerate(sets):
            n = int(cnt[s])
            if n == 0:
                continue
            want3, wantm = refs[s].stereo_depth(d["img0"], d["img1"], p2d[s, :n], p2u[s, :n], p3w[s, :n], has[s, :n], poses[s], rng)
            assert np.array_equal(gotm[s, :n], wantm), (call, s, np.flatnonzero(gotm[s, :n] != wantm)[:8])
            assert np.array_equal(got3[s, :n], want3), (call, s, np.abs(got3[s, :n] - want3).max())
            seen_fail += int((wantm == 0).sum())
            seen_ok += int((wantm == 1).sum())
            z = want3[wantm == 0, 2]
            assert np.all((z >= 0.3) & (z < 0.7000001))          # rand()-drawn dummy depths
    assert seen_ok > 200 and seen_fail > 20, (seen_ok, seen_fail)  # both branches exercised
    with pytest.raises(flvis_amd.FlvisError):                    # a depth-camera rig has no stereo pair
        dcfg, _ = _cfgs(synth.D435I_DEPTH_YAML, "depth")
        ctx.stereo_depth(dcfg, dev(img0), dev(img1), dev(p2d), dev(p2u), dev(p3w), dev(has), dev(cnt), poses, 3.0, state)
    with pytest.raises(flvis_amd.FlvisError):                    # a configuration that was never finalised: empty projections, refused
        import copy
        raw = copy.copy(cfg)
        for k in range(12):
            raw.P0[k] = 0.0
            raw.P1[k] = 0.0
        ctx.stereo_depth(raw, dev(img0), dev(img1), dev(p2d), dev(p2u), dev(p3w), dev(has), dev(cnt), poses, 3.0, state)
    ctx.close()
