"""GPU tests at the sizes BASELINE.json's configs name (through the C ABI, against the CPU oracle):

  configs[3]  one GPU, a batch of 64 independent 640x480 stereo+IMU streams, front-end + batched Schur BA
  configs[2]  one stream, full HIP front-end + HIP sliding-window BA together (EuRoC-like rig: window 10, <= 480 landmarks)

The kernel-level and two-stream closed-loop parity tests live in test_gpu_image.py / test_gpu_pipeline.py; these two check
that nothing changes at the batch size the benchmark runs at, that a stream's result does not depend on the batch it is in,
and that the local map fed by the device-side keyframe queue produces what the reference's callback produces."""
import ctypes as C
import hashlib
import os
import tempfile

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


def _cfgs_yaml(text, tag):
    import flvis_amd
    p = os.path.join(tempfile.gettempdir(), "flvis_cfgtest_%s.yaml" % tag)
    open(p, "w").write(text)
    cfg = flvis_amd.load_config(p)
    ocfg = O.RefConfig()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg))
    return cfg, ocfg


def _run_batch(cfg, S, nframes, sampled, ocfg=None, lanes=None, input_hold=0):
    """Feeds `nframes` frames of S synthetic streams with the local map on.  Returns per-frame outputs of the sampled streams,
    all trajectories, the counters and (when ocfg is given) the oracle's per-frame results for the sampled streams."""
    import flvis_amd
    import torch
    from flvis_amd import synth
    if lanes is not None:
        os.environ["FLVIS_LANES"] = str(lanes)
    try:
        ctx = flvis_amd.Context(0)
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
    finally:
        os.environ.pop("FLVIS_LANES", None)
    held = []            # with input_hold = n the caller leaves a frame's input buffers alone during the next n calls
    if input_hold:
        trk.set_input_hold(input_hold)
    skip = cfg.skip_first_n_imgs
    trajs = [synth.Trajectory(s) for s in range(S)]
    rnd = synth.Renderer("cuda")
    refs = {i: O.Tracker(ocfg, 0xF1715 + i) for i in sampled} if ocfg is not None else {}
    lmaps = {i: O.LocalMap(cfg.window_size, np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])) for i in refs}
    last_corr = {i: None for i in refs}
    got, want = {i: [] for i in sampled}, {i: [] for i in sampled}
    t_prev = -1.0 / synth.FRAME_HZ
    standin = None
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i in range(S):
            smp = synth.imu_samples(trajs[i], i, t_prev, t)
            trk.imu_feed_flvis(i, smp)
            if i in refs:
                for r in smp:
                    refs[i].imu(r[0], r[1:4], r[4:7])
        t_prev = t
        if f >= skip or standin is None:   # the skipped start-up frames are never looked at
            i0, i1 = rnd.stereo_frame(trajs, t, f)
            if standin is None:
                standin = (i0, i1)
        else:
            i0, i1 = standin
        if input_hold:   # free-running lanes: no per-frame read-back (it would wait for every lane), only the recorded trajectory
            held = (held + [(i0, i1)])[-(input_hold + 1):]
            trk.image_feed(i0, i1, [t] * S, want_out=False, with_local_map=True)
            continue
        outs = trk.image_feed(i0, i1, [t] * S, with_local_map=True)
        for i in sampled:
            got[i].append(outs[i])
        for i in refs:
            w = refs[i].image(t, i0[i].cpu().numpy(), i1[i].cpu().numpy())
            if w["new_keyframe"]:
                kf = refs[i].keyframe()
                c = lmaps[i].push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
                if c is not None:
                    last_corr[i] = c
            want[i].append(w)
    rows = np.stack([trk.trajectory(i, 0, nframes) for i in range(S)])
    counters = trk.counters()
    corr = {i: trk.correction(i) for i in sampled}
    dbg = (C.c_int64 * 64)()
    ctx._check(ctx._lib.flvis_debug_counters(ctx._h, dbg), "debug_counters")
    lanes_used = ctx._lib.flvis_tracker_lanes(ctx._h)
    ctx.close()
    torch.cuda.synchronize()
    return dict(got=got, want=want, rows=rows, counters=counters, corr=corr, ref_corr=last_corr, dropped=int(dbg[3]), lanes=lanes_used)


def _compare_stream(got, want, where):
    """Frame-by-frame comparison of one stream: LOCKSTEP for the whole run -- every discrete decision and the pose itself are
    identical on both sides (see tests/test_gpu_pipeline.py for what makes that possible).  Returns the tracked frames compared."""
    n = 0
    for f, (g, w) in enumerate(zip(got, want)):
        tag = "%s frame %d" % (where, f)
        assert g["state"] == w["state"] and g["new_keyframe"] == w["new_keyframe"], tag
        assert g["n_landmarks"] == w["n_landmarks"] and np.array_equal(g["dbg"], w["dbg"]), tag
        assert np.array_equal(g["pose7"], w["pose7"]), (tag, g["pose7"] - w["pose7"])
        n += w["state"] == 1
    return n


def test_config3_batch_of_64_streams_with_local_map():
    """BASELINE configs[3]: S = 64, the reference's 50 skipped start-up frames + 22 processed frames, local map on.
    * every stream reaches the Tracking state and stays there; keyframes flow into the local map and none is dropped;
    * streams 0, 31 and 63 agree with the CPU oracle frame by frame (and their final CorrectionInf names the same keyframe);
    * the run is deterministic: a second run gives bit-identical trajectories for all 64 streams;
    * a stream's result does not depend on the batch / lane partition it runs in (2 lanes vs 1)."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.D435I_STEREO_YAML, "d435_stereo")
    S, nframes = 64, cfg.skip_first_n_imgs + 22
    sampled = [0, 31, 63]
    a = _run_batch(cfg, S, nframes, sampled, ocfg)
    assert a["lanes"] == 1 and a["dropped"] == 0
    states = a["rows"][:, :, 8].astype(int) & 15
    assert np.all(states[:, :cfg.skip_first_n_imgs] == 0)                       # skipped frames: nothing is processed
    assert np.all(states[:, cfg.skip_first_n_imgs:] == 1), "a stream left the Tracking state"
    assert a["counters"][0] == S * nframes and a["counters"][1] >= 4 * S        # >= 4 keyframes per stream in 22 frames
    for i in sampled:
        assert _compare_stream(a["got"][i], a["want"][i], "stream %d" % i) == 22
        gc, wc = a["corr"][i], a["ref_corr"][i]                                 # the last CorrectionInf of the stream
        assert (gc is None) == (wc is None)                                     # (22 frames rarely fill a window: the 40-frame test below compares the optimiser)
        if gc is not None:      # same keyframes in (bit-identical front-ends), same bookkeeping; the optimiser sums in another order
            assert gc["frame_id"] == wc["frame_id"]
            assert np.array_equal(gc["lm_id"], wc["lm_id"]) and np.array_equal(gc["outlier_id"], wc["outlier_id"])
            assert np.allclose(gc["pose7"], wc["pose7"], atol=1e-6, rtol=0)
            assert np.allclose(gc["lm_3d"], wc["lm_3d"], atol=1e-6, rtol=0)
    digest = hashlib.sha256(a["rows"].tobytes()).hexdigest()
    b = _run_batch(cfg, S, nframes, sampled)
    assert hashlib.sha256(b["rows"].tobytes()).hexdigest() == digest, "two identical runs differ"
    assert b["counters"] == a["counters"]
    c = _run_batch(cfg, S, nframes, sampled, lanes=2)
    assert c["lanes"] == 2 and np.array_equal(c["rows"], a["rows"]), "a stream's trajectory depends on the lane partition"
    d = _run_batch(cfg, S, nframes, [], lanes=4, input_hold=3)   # lanes out of step by up to 3 frames (flvis_set_input_hold)
    assert d["lanes"] == 4 and d["dropped"] == 0 and np.array_equal(d["rows"], a["rows"]), "free-running lanes change a trajectory"


def test_config3_batch_of_64_streams_local_map_optimises():
    """The same batch, long enough for every window (8 keyframes) to fill: the batched Schur BA runs for all 64 streams, no
    keyframe is dropped, the corrected keyframe pose stays close to the tracker's own pose of that frame."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.D435I_STEREO_YAML, "d435_stereo")
    S, nframes = 64, cfg.skip_first_n_imgs + 40
    a = _run_batch(cfg, S, nframes, [0, 63], ocfg)
    states = a["rows"][:, :, 8].astype(int) & 15
    assert np.all(states[:, cfg.skip_first_n_imgs:] == 1) and a["dropped"] == 0
    assert a["counters"][2] >= S, a["counters"]                                 # at least one optimisation per stream
    for i in (0, 63):
        c = a["corr"][i]
        assert c is not None and len(c["lm_id"]) > 100
        fid = c["frame_id"]                                                     # frame ids count image_feed calls from 1
        assert np.abs(a["rows"][i, fid - 1, 1:4] - c["pose7"][:3]).max() < 0.05
        # configs[3]'s optimiser against the oracle AT S = 64 (not only the one-stream tests): the front-ends are in lockstep for all
        # 40 frames, so both local maps get the same keyframes; the window has optimised on both sides, and the last CorrectionInf names
        # the same keyframe, landmarks and outliers (exact) with the values of the fp64 LM chain within 1e-6
        assert _compare_stream(a["got"][i], a["want"][i], "stream %d" % i) == 40
        wc = a["ref_corr"][i]
        assert wc is not None, "the oracle's window of stream %d never optimised: nothing was compared" % i
        assert c["frame_id"] == wc["frame_id"]
        assert np.array_equal(c["lm_id"], wc["lm_id"]) and np.array_equal(c["outlier_id"], wc["outlier_id"])
        assert np.allclose(c["pose7"], wc["pose7"], atol=1e-6, rtol=0), c["pose7"] - wc["pose7"]
        assert np.allclose(c["lm_3d"], wc["lm_3d"], atol=1e-6, rtol=0), np.abs(c["lm_3d"] - wc["lm_3d"]).max()


def test_config2_single_stream_frontend_and_ba_together():
    _config2_frontend_and_ba(imu_factor=False)


def test_config2_with_imu_rotation_factor():
    """The same loop with the IMU rotation factor enabled on both sides (flvis_set_imu_factor): the gyro
    preintegration the front-end attaches to every keyframe reaches the window BA through the keyframe queue."""
    _config2_frontend_and_ba(imu_factor=True)


def test_config2_with_the_full_imu_factor():
    """... and with the factor's position rows (flvis_set_imu_factor_accel): the displacement preintegrated by the front-end and the
    filter velocity of the previous keyframe travel with every keyframe through the queue into the 6-row edges of the window BA."""
    _config2_frontend_and_ba(imu_factor=True, sigma_a=0.1)


def _config2_frontend_and_ba(imu_factor, sigma_a=0.0):
    """BASELINE configs[2] on the EuRoC-like rig (window 10, <= 480 landmarks per frame): ONE stream through the HIP front-end
    with the HIP local map consuming its keyframe queue, beside the oracle front-end feeding the oracle's LocalMap.  After
    every keyframe the CorrectionInf of both sides is compared: same keyframe, same landmark id list, same outliers while the
    front-ends are in lockstep; poses and landmarks within the tolerances below."""
    import flvis_amd
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    assert cfg.window_size == 10
    rig = synth.euroc_rig()
    nframes, sid = 64, 9
    ctx = flvis_amd.Context(0)
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=0xF1715, traj_capacity=nframes)
    ref = O.Tracker(ocfg, 0xF1715)
    lmap = O.LocalMap(cfg.window_size, np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]]))
    n_links = 0
    if imu_factor:
        import _geom as G
        sigma_g = 0.004
        R_c_i = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)[:3, :3].T
        q = G.pose7(R_c_i, np.zeros(3))
        lmap.set_imu_factor(True, sigma_g, np.array([q[6], q[3], q[4], q[5]]))
        trk.set_imu_factor(True, sigma_g)
        if sigma_a > 0:
            T_i_c = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
            lmap.set_imu_factor_pos(sigma_a, -R_c_i @ T_i_c[:3, 3])        # body origin in the camera frame (T_c_i's translation)
            trk.set_imu_factor_accel(sigma_a)
    tr = synth.Trajectory(sid)
    rnd = synth.Renderer("cuda", rig=rig)
    t_prev = -0.05
    n_corr, max_lm = 0, 0
    want_c = None
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        smp = synth.imu_samples(tr, sid, t_prev, t)
        trk.imu_feed_flvis(0, smp)
        for r in smp:
            ref.imu(r[0], r[1:4], r[4:7])
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, f)
        g = trk.image_feed(i0, i1, [t], with_local_map=True)[0]
        w = ref.image(t, i0[0].cpu().numpy(), i1[0].cpu().numpy())
        where = "frame %d" % f
        assert g["state"] == w["state"] and g["new_keyframe"] == w["new_keyframe"], where
        assert g["n_landmarks"] == w["n_landmarks"] and np.array_equal(g["dbg"], w["dbg"]), where
        assert np.array_equal(g["pose7"], w["pose7"]), (where, g["pose7"] - w["pose7"])      # front-end: lockstep
        max_lm = max(max_lm, g["n_landmarks"])
        if not w["new_keyframe"]:
            continue
        kf = ref.keyframe()
        valid, dq, dt = ref.keyframe_imu()
        if imu_factor and valid:
            lmap.next_imu(dq, dt)
            if sigma_a > 0:
                lmap.next_imu_pos(*ref.keyframe_imu_pos())
            n_links += 1
        c = lmap.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        if c is not None:
            want_c = c
        got_c = trk.correction(0)                      # drains the stream's keyframe queue first
        assert (got_c is None) == (want_c is None), where
        if want_c is None:
            continue
        n_corr += 1
        # identical keyframes went in: the bookkeeping (ids, outliers) is exact, the optimiser (20 LM iterations, its sums are
        # register-tile / wave reductions in another order than the restatement's loops) agrees to 1e-6 m
        assert got_c["frame_id"] == want_c["frame_id"], where
        assert np.array_equal(got_c["lm_id"], want_c["lm_id"]), where
        assert np.array_equal(got_c["outlier_id"], want_c["outlier_id"]), where
        assert np.allclose(got_c["pose7"], want_c["pose7"], atol=1e-6, rtol=0), (where, got_c["pose7"] - want_c["pose7"])
        assert np.allclose(got_c["lm_3d"], want_c["lm_3d"], atol=1e-6, rtol=0), (where, np.abs(got_c["lm_3d"] - want_c["lm_3d"]).max())
    ctx.close()
    assert n_corr >= 3, n_corr
    assert 250 <= max_lm <= 480, max_lm
    assert n_links >= 10 if imu_factor else n_links == 0
