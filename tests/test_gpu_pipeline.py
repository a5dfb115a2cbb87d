"""GPU parity tests of the full hot path through the C ABI: batched HIP front-end (F2FTracking mirror) and sliding-window
BA (LocalMap mirror) against the CPU oracle on identical synthetic inputs.
Bar: landmark ids / counts / inlier flags / tracked float pixel positions bit-exact; fp64 geometry within the tolerance
written next to each check."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import _ba_synth as B
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import flvis_amd
    c = flvis_amd.Context(0)
    yield c
    c.close()


def _cfgs():
    import flvis_amd
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_d435_stereo.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    cfg = flvis_amd.load_config(p)
    ocfg = O.RefConfig()
    assert C.sizeof(ocfg) == C.sizeof(cfg)
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg))  # identical layout: both sides get the SAME numbers
    return cfg, ocfg


def _cfgs_yaml(text, tag):
    import flvis_amd
    p = os.path.join(tempfile.gettempdir(), "flvis_test_%s.yaml" % tag)
    open(p, "w").write(text)
    cfg = flvis_amd.load_config(p)
    ocfg = O.RefConfig()
    assert C.sizeof(ocfg) == C.sizeof(cfg)
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(cfg))  # identical layout: both sides get the SAME numbers
    return cfg, ocfg


class _HostFeeder:
    """The caller's side of flvis_image_feed_host (what image_input_callback holds: two cv::Mat per frame, vo_tracking.cpp:396-430):
    host buffers in the layout a test asks for, handed over and then deliberately overwritten as soon as the contract allows it --
    right after the call with hold_buffers = 0, after the NEXT call has returned with hold_buffers = 1 -- so that an upload still in
    flight when the buffer is reused, or a frame that reads the staging slot of its successor, breaks the lockstep comparison.
    layout: "block" (one [S][H][W*c] block: a single copy), "separate" (one allocation per image), "padded" (rows padded by 64 B);
    pinned: page-locked memory (truly asynchronous uploads) or pageable."""

    def __init__(self, S, channels=1, hold=0, layout="block", pinned=True, scribble=True):
        self.S, self.ch, self.hold, self.layout, self.pinned, self.scribble = S, channels, hold, layout, pinned, scribble
        self.sets = {}
        self.n = 0
        self.prev = None
        self.rng = np.random.default_rng(99)

    def _alloc(self, nbytes):
        import torch
        t = torch.empty(nbytes, dtype=torch.uint8, pin_memory=self.pinned)
        return t, t.numpy()

    def _views(self, key, shape, dtype):
        """S per-stream views [H, W(, c)] of this set's buffers (allocated once per set and camera)."""
        if key not in self.sets:
            H, W = shape[0], shape[1]
            c = shape[2] if len(shape) == 3 else 1
            row = W * c * np.dtype(dtype).itemsize
            pitch = row + (64 if self.layout == "padded" else 0)
            keep, views = [], []
            if self.layout == "separate":
                for s in range(self.S):
                    t, a = self._alloc(H * pitch + 4096 * (s % 3))      # (different sizes: never one contiguous block)
                    keep.append(t)
                    views.append(a[:H * pitch])
            else:
                t, a = self._alloc(self.S * H * pitch)
                keep.append(t)
                views = [a[s * H * pitch:(s + 1) * H * pitch] for s in range(self.S)]
            out = []
            for v in views:
                v = v.reshape(H, pitch)[:, :row]
                v = v.view(dtype)
                out.append(v.reshape(H, W, c) if c > 1 else v.reshape(H, W))
            self.sets[key] = (keep, out)
        return self.sets[key][1]

    def colour(self, gray):
        """A BGR(A) image whose channels differ (so that the conversion's weights matter), from a rendered mono image."""
        g = gray.astype(np.int16)
        d = self.rng.integers(-12, 13, size=gray.shape + (3,), dtype=np.int16)
        bgr = np.clip(g[..., None] + d, 0, 255).astype(np.uint8)
        if self.ch == 4:
            bgr = np.concatenate([bgr, self.rng.integers(0, 256, size=gray.shape + (1,), dtype=np.uint8)], axis=-1)
        return bgr

    def feed(self, trk, h0, h1, times, depth, **kw):
        """h0 / h1: [S, H, W] mono (h1 uint16 on depth rigs).  Returns (outs, the mono images the tracker must have worked on)."""
        k = self.n % 2 if self.hold else 0
        src0 = [self.colour(h0[s]) for s in range(self.S)] if self.ch > 1 else list(h0)
        src1 = [self.colour(h1[s]) for s in range(self.S)] if (self.ch > 1 and not depth) else list(h1)
        v0 = self._views((k, 0), src0[0].shape, src0[0].dtype)
        v1 = self._views((k, 1), src1[0].shape, src1[0].dtype)
        for s in range(self.S):
            v0[s][...] = src0[s]
            v1[s][...] = src1[s]
        outs = trk.image_feed_host(v0, v1, times, hold_buffers=bool(self.hold), **kw)
        if self.scribble:
            if not self.hold:
                for v in v0 + v1:
                    v[...] = 0xA5        # "the caller may reuse its buffers immediately"
            elif self.prev is not None:
                for v in self.prev:
                    v[...] = 0x5A        # the previous call's buffers are free now that this call has returned
        self.prev = v0 + v1
        self.n += 1
        g0 = np.stack([O.cvt_bgr_to_gray(x) for x in src0]) if self.ch > 1 else h0
        g1 = np.stack([O.cvt_bgr_to_gray(x) for x in src1]) if (self.ch > 1 and not depth) else h1
        return outs, g0, g1


def _run_frontend_parity(ctx, cfg, ocfg, rig, streams, nframes, min_lock, min_kf, depth_range=None, imu=True, t_offset=0.0, host=None,
                         check=None):
    import flvis_amd
    from flvis_amd import synth
    S = len(streams)
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda", rig=rig)
    seed_base = 0xF1715
    trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=seed_base, traj_capacity=nframes)
    check = list(range(S)) if check is None else list(check)     # the streams run beside the oracle (all of them by default)
    refs = {i: O.Tracker(ocfg, seed_base + i) for i in check}
    feeder = _HostFeeder(S, **host) if host is not None else None
    t_prev = -0.05
    n_kf = n_imu_links = n_imu_rows = 0
    lock_frames = [0] * S      # tracked frames (all compared exactly)
    imu_want = [[] for _ in range(S)]   # F2FTracking::imu_feed's outputs per sample (q_w_i, pos_w_i, vel_w_i) on the oracle side
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i, s in enumerate(streams):
            if not imu:
                break                                  # rigs without an IMU (type_of_vi 4): no sample is ever fed
            smp = synth.imu_samples(trajs[i], s, t_prev, t)
            smp[:, 0] += t_offset                      # stamps as a dataset carries them (EuRoC: seconds since 1970)
            trk.imu_feed_flvis(i, smp)
            if i in refs:
                for r in smp:
                    imu_want[i].append(np.concatenate([[r[0]], refs[i].imu(r[0], r[1:4], r[4:7])]))
        t_prev = t
        if depth_range is None:
            i0, i1 = rnd.stereo_frame(trajs, t, f)
            h0, h1 = i0.cpu().numpy(), i1.cpu().numpy()
        else:  # depth-camera mode: the second image is the Z16 depth image aligned to cam0
            i0, i1 = rnd.depth_frame(trajs, t, f, max_range=depth_range)
            h0, h1 = i0.cpu().numpy(), i1.cpu().numpy().view(np.uint16)
        if feeder is None:
            outs = trk.image_feed(i0, i1, [t + t_offset] * S, with_local_map=False)
        else:   # the nodelet's entry: host images in (and, for colour input, the mono images cvtColor makes of them for the oracle)
            outs, h0, h1 = feeder.feed(trk, h0, h1, [t + t_offset] * S, depth_range is not None, with_local_map=False)
        for i in check:
            # the IMU-rate trajectory (/imu_pose, what the reference records on EuRoC): every sample's q / p / v bit-identical;
            # fetched every third frame so that a fetch spans several image feeds (and the vision corrections between them)
            if imu and (f % 3 == 2 or f == nframes - 1):
                rows, dropped = trk.imu_states(i)
                assert dropped == 0 and len(rows) == len(imu_want[i]), ("frame %d stream %d" % (f, i), len(rows), len(imu_want[i]))
                assert np.array_equal(rows, np.array(imu_want[i]).reshape(-1, 11)), "IMU states, frame %d stream %d" % (f, i)
                n_imu_rows += len(rows)
                imu_want[i] = []
            want = refs[i].image(t + t_offset, h0[i], h1[i])
            got = outs[i]
            where = "frame %d stream %d" % (f, i)
            # LOCKSTEP for the whole run: every discrete decision and every fp64 value of the frame is IDENTICAL on both sides
            assert got["state"] == want["state"] and got["new_keyframe"] == want["new_keyframe"], where
            assert got["n_landmarks"] == want["n_landmarks"], (where, got["n_landmarks"], want["n_landmarks"])
            assert np.array_equal(got["dbg"], want["dbg"]), (where, got["dbg"], want["dbg"])       # LK / F / PnP inlier counts
            assert np.array_equal(got["pose7"], want["pose7"]), (where, got["pose7"] - want["pose7"])  # bit-identical pose
            if want["state"] == 1:
                lock_frames[i] += 1
                gl, wl = trk.landmarks(i), refs[i].landmarks()
                assert np.array_equal(gl["ids"], wl["ids"]) and np.array_equal(gl["flags"], wl["flags"]), where
                assert np.array_equal(gl["p2d"], wl["p2d"]) and np.array_equal(gl["p2u"], wl["p2u"]), where
                assert np.array_equal(gl["p3w"], wl["p3w"]), (where, np.abs(gl["p3w"] - wl["p3w"]).max())
            if want["new_keyframe"]:
                n_kf += 1
                gk, wk = trk.keyframe(i), refs[i].keyframe()
                assert gk["frame_id"] == wk["frame_id"] and np.array_equal(gk["lm_id"], wk["lm_id"]), where
                assert np.array_equal(gk["lm_2d"], wk["lm_2d"]) and np.array_equal(gk["lm_3d"], wk["lm_3d"]), where
                assert np.array_equal(gk["pose7"], wk["pose7"]), where
                gv, gdq, gdt = trk.get_keyframe_imu(i)             # gyro preintegration since the previous keyframe
                wv, wdq, wdt = refs[i].keyframe_imu()
                assert gv == wv and gdt == wdt and np.array_equal(gdq, wdq), (where, gdq - wdq, gdt - wdt)
                gdp, gva = trk.get_keyframe_imu_pos(i)            # ... and the position part (displacement, previous keyframe's velocity)
                wdp, wva = refs[i].keyframe_imu_pos()
                assert np.array_equal(gdp, wdp) and np.array_equal(gva, wva), (where, gdp - wdp, gva - wva)
                n_imu_links += int(wv)
    assert n_kf >= min_kf
    assert (n_imu_rows >= 9 * len(check) * (nframes - 1)) if imu else (n_imu_rows == 0)
    assert (n_imu_links >= (min_kf - len(check)) // 2) if imu else (n_imu_links == 0)
    assert min(lock_frames[i] for i in check) >= min_lock, lock_frames
    rows = trk.trajectory(0, 0, nframes)
    assert np.allclose(rows[:, 0], np.arange(nframes) / synth.FRAME_HZ + t_offset, rtol=0, atol=1e-6)




def test_frontend_parity_two_streams(ctx):
    """Closed-loop parity of the whole front-end against the oracle, two streams, 100 frames (50 of them tracked), in
    LOCKSTEP for the whole run: state, keyframe flag, landmark count, LK / F / PnP inlier counts, landmark ids, flags, pixel
    positions, 3-D points, keyframe payloads and the pose itself are bit-identical every frame.  What makes that possible:
    both sides execute the same elementary functions (csrc/det_math.hpp), every fp64 sum on the path is a sequential sum in
    the order the reference's loops / g2o's active-edge list define (k_pose_lm, the PnP refinement, the reprojection mean),
    the 6x6 solves read the upper triangle (as g2o's SimplicialLDLT<Upper> does), and the host-side rig transforms are
    inverted with Sophus' formulas on both sides.  ("feature indices bit-exact" of BASELINE.json, for the path.)"""
    from flvis_amd import synth
    cfg, ocfg = _cfgs()
    _run_frontend_parity(ctx, cfg, ocfg, None, [3, 140], 100, 49, 4)


def test_frontend_parity_euroc_mode(ctx):
    """Same closed-loop comparison in EuRoC mode (type_of_vi 1): 752x480, equalizeHist on both images, unrectified stereo
    with radial-tangential distortion (undistortPoints + stereoRectify'd projection matrices), no skipped frames, the
    EuRoC feature parameters (1000 corners, quality 0.01, minDistance 10)."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    assert cfg.cam_type == 1 and cfg.need_equal_hist == 1 and cfg.image_width == 752
    _run_frontend_parity(ctx, cfg, ocfg, synth.euroc_rig(), [9], 60, 50, 2)


def test_frontend_parity_euroc_mode_epoch_stamps(ctx):
    """The EuRoC-mode comparison with the stamps a dataset carries (seconds since 1970, ~1.4e9: a double resolves 2.4e-7 s there):
    every difference of two stamps (IMU dt, the vision correction's dt, the state lookup) must round the same way on both sides."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    _run_frontend_parity(ctx, cfg, ocfg, synth.euroc_rig(), [9], 40, 30, 2, t_offset=1403636579.0)


def test_frontend_parity_depth_camera_mode(ctx):
    """SURVEY 8f-4 (first half): DEPTH_D435 (type_of_vi 0).  Same closed-loop comparison; the depth of a landmark comes from
    the nearest pixel of the Z16 image (camera_frame.cpp:182-234) instead of stereo LK, pixels beyond 3.3 m carry no
    depth (0) so the rand()-dummy branch and the "no measurement at all" branch are exercised, LK guesses go through the
    pinhole projection (lkorb_tracking.cpp:41-52), nothing is undistorted."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.D435I_DEPTH_YAML, "d435i_depth")
    assert cfg.cam_type == 2 and cfg.depth_factor == 1000.0 and cfg.skip_first_n_imgs == 50
    _run_frontend_parity(ctx, cfg, ocfg, None, [3, 77], 50 + 36, 35, 3, depth_range=3.3)


def test_frontend_parity_with_opencvs_iterative_tail(ctx, monkeypatch):
    """FLVIS_PNP_TAIL=cv: behind the PnP RANSAC the device runs the final solve of solvePnPRansac(ITERATIVE) the way OpenCV does -- a DLT
    start and CvLevMarq on the inliers (k_pnp_tail_cv: cv_solvers.hpp's find_extrinsic_iterative, one wave per stream) -- instead of the
    Gauss-Newton refinement of the winning model.  Against the checker built the same way (`make -C oracle TAIL=cv`) the closed loop is in
    LOCKSTEP again, bit for bit, on the D435 rig (two streams, 100 frames) and the EuRoC-like one; the two tails differ from each other
    from the first tracked frame on (4.4e-9 m, tests/test_oracle_tracking.py), so a device that still ran the default tail would fail
    here, and the default tests would fail with this one."""
    from flvis_amd import synth
    monkeypatch.setenv("FLVIS_PNP_TAIL", "cv")
    O.use_sum_order("tail_cv")
    try:
        cfg, ocfg = _cfgs()
        _run_frontend_parity(ctx, cfg, ocfg, None, [3, 140], 100, 49, 4)
        cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
        _run_frontend_parity(ctx, cfg, ocfg, synth.euroc_rig(), [9], 60, 50, 2)
    finally:
        O.use_sum_order("product")


def test_frontend_parity_kitti_mode(ctx):
    """type_of_vi 4 (vo_tracking.cpp:146,265-306): rectified stereo from two projection matrices, no IMU (fixed initial
    attitude, P3P without a prior), no skipped frames, 1241 x 376 tightly packed images (rows not dword aligned: both images
    are copied into pitch-aligned pyramids), GFTT with 2 x 2000 corners.  Same closed-loop comparison against the oracle."""
    import flvis_amd
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.KITTI_LIKE_YAML, "kitti_like")
    assert (cfg.type_of_vi, cfg.cam_type, cfg.imu_type, cfg.skip_first_n_imgs, cfg.image_width) == (4, 0, 3, 0, 1241)
    _run_frontend_parity(ctx, cfg, ocfg, synth.kitti_like_rig(), [5, 77], 26, 26, 3, imu=False)
    trk = flvis_amd.Tracker(ctx, cfg, 1)
    with pytest.raises(flvis_amd.FlvisError):          # imu_callback has no remap for imu_type NONE
        trk.imu_feed_sensor(0, 0.0, [0, 0, 9.81], [0, 0, 0])


# ---- the boundary's real entry: flvis_image_feed_host (what ros/src/tracking_nodelet.cpp and INTEGRATION.md call) in lockstep ----

def test_host_feed_parity_two_streams_buffers_reused_at_once(ctx):
    """The two-stream lockstep run through flvis_image_feed_host: page-locked mono8 images in one block, hold_buffers = 0, and the
    caller overwrites its buffers the moment the call returns.  Same asserts as test_frontend_parity_two_streams (state, keyframe
    flag, ids, flags, pixels, 3-D points, fp64 pose bit-identical, every frame)."""
    cfg, ocfg = _cfgs()
    _run_frontend_parity(ctx, cfg, ocfg, None, [3, 140], 100, 49, 4, host=dict(channels=1, hold=0, layout="block", pinned=True))


def test_host_feed_parity_batch_of_64_held_buffers(ctx):
    """S = 64 (BASELINE configs[3]'s batch) through the host entry with hold_buffers = 1: the call returns while the 39 MB of uploads
    are still in flight, the caller alternates two buffer sets and overwrites a set only after the NEXT call has returned.  Streams
    0, 31 and 63 run beside the oracle in lockstep."""
    cfg, ocfg = _cfgs()
    streams = [3 + 7 * i for i in range(64)]
    _run_frontend_parity(ctx, cfg, ocfg, None, streams, 50 + 30, 29, 6, host=dict(channels=1, hold=1, layout="block", pinned=True),
                         check=[0, 31, 63])


def test_host_feed_parity_batch_of_64_pageable_reused_at_once(ctx):
    """The same batch from pageable memory in one allocation per image (64 + 64 copies per frame), hold_buffers = 0, overwritten at once."""
    cfg, ocfg = _cfgs()
    streams = [3 + 7 * i for i in range(64)]
    _run_frontend_parity(ctx, cfg, ocfg, None, streams, 50 + 16, 15, 3, host=dict(channels=1, hold=0, layout="separate", pinned=False),
                         check=[0, 63])


def test_host_feed_parity_euroc_mode_bgr_padded_rows(ctx):
    """EuRoC mode (752 x 480, equalizeHist) from 3-channel BGR images with padded rows in pageable memory: the conversion
    (cv::cvtColor at f2f_tracking.cpp:74-111) runs on the device, the oracle gets the restated conversion's mono images."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    _run_frontend_parity(ctx, cfg, ocfg, synth.euroc_rig(), [9], 60, 50, 2, host=dict(channels=3, hold=0, layout="padded", pinned=False))


def test_host_feed_parity_depth_camera_mode_bgra_and_z16(ctx):
    """Depth rig: img0 as BGRA, img1 the 16UC1 depth image (2 bytes per pixel), separate allocations, held buffers."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.D435I_DEPTH_YAML, "d435i_depth")
    _run_frontend_parity(ctx, cfg, ocfg, None, [3, 77], 50 + 36, 35, 3, depth_range=3.3,
                         host=dict(channels=4, hold=1, layout="separate", pinned=True))


def test_host_feed_parity_kitti_mode_unaligned_rows(ctx):
    """KITTI-like rig: 1241-pixel rows (not dword aligned) from page-locked memory, padded, reused at once."""
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.KITTI_LIKE_YAML, "kitti_like")
    _run_frontend_parity(ctx, cfg, ocfg, synth.kitti_like_rig(), [5, 77], 26, 26, 3, imu=False,
                         host=dict(channels=1, hold=0, layout="padded", pinned=True))


@pytest.mark.parametrize("hold", [0, 1])
def test_host_feed_without_readback_equals_the_resident_run(ctx, hold):
    """The throughput form of the host entry (h_out = NULL: the call never synchronises, so uploads, staging slots and frames really
    overlap -- bench.py's with_h2d leg) with the local map on, buffers overwritten as early as the contract allows: the whole
    trajectory of every stream, the last landmarks and the last CorrectionInf are bit-identical to a run of the same frames through
    flvis_image_feed with device-resident images.  An upload racing the previous frame's ingest, or a staging slot refilled before
    its frame has consumed it, shows here."""
    import flvis_amd
    from flvis_amd import synth
    cfg, _ = _cfgs()
    S, nframes = 64, 50 + 30
    streams = [5 + 3 * i for i in range(S)]
    res = []
    for mode in ("resident", "host"):
        trajs = [synth.Trajectory(s) for s in streams]
        rnd = synth.Renderer("cuda")
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
        feeder = _HostFeeder(S, channels=1, hold=hold, layout="block", pinned=True)
        t_prev = -0.05
        for f in range(nframes):
            t = f / synth.FRAME_HZ
            for i, s in enumerate(streams):
                trk.imu_feed_flvis(i, synth.imu_samples(trajs[i], s, t_prev, t))
            t_prev = t
            i0, i1 = rnd.stereo_frame(trajs, t, f)
            if mode == "resident":
                trk.image_feed(i0, i1, [t] * S, want_out=False, with_local_map=True)
                ctx.synchronize()           # (the renderer reuses nothing, but keep the resident run simple: one frame at a time)
            else:
                feeder.feed(trk, i0.cpu().numpy(), i1.cpu().numpy(), [t] * S, False, want_out=False, with_local_map=True)
        rows = np.stack([trk.trajectory(i, 0, nframes) for i in range(S)])
        lms = [trk.landmarks(i) for i in (0, S - 1)]
        corr = [trk.correction(i) for i in (0, S - 1)]
        cnt = trk.counters()
        res.append((rows, lms, corr, cnt))
        del trk
    (ra, la, ca, na), (rb, lb, cb, nb) = res
    assert np.all((ra[:, 50:, 8].astype(int) & 15) == 1)                        # every stream tracked after the skipped frames
    assert np.array_equal(ra, rb), np.abs(ra - rb).max()
    for x, y in zip(la, lb):
        for k in ("ids", "flags", "p2d", "p2u", "p3w"):
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(ca, cb):
        assert x is not None and y is not None and x["frame_id"] == y["frame_id"]
        assert np.array_equal(x["lm_id"], y["lm_id"]) and np.array_equal(x["outlier_id"], y["outlier_id"])
        assert np.array_equal(x["pose7"], y["pose7"]) and np.array_equal(x["lm_3d"], y["lm_3d"])
    assert list(na) == list(nb), (na, nb)


def test_run_steps_batches_equal_frame_by_frame_feeds(ctx):
    """flvis_run_steps (what bench.py times) against flvis_imu_feed_all + flvis_image_feed frame by frame, local map on: between the
    steps of a batch the local-map launch of a step is enqueued inside the NEXT step (FLVIS_BA_START, pipeline.cpp), the last step of
    a batch launches at its end -- batches of 1, 2 and many steps, back to back without a synchronise, must leave the same
    trajectories, landmarks, CorrectionInf and counters, bit for bit."""
    import flvis_amd
    from flvis_amd import synth
    cfg, _ = _cfgs()
    S, nframes = 16, 50 + 48
    streams = [7 + 5 * i for i in range(S)]
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda")
    frames, t_prev = [], -0.05
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        smp = [synth.imu_samples(trajs[i], s, t_prev, t) for i, s in enumerate(streams)]
        t_prev = t
        n = max(len(x) for x in smp)
        cnt = np.array([len(x) for x in smp], np.int32)
        blk = np.zeros((S, max(n, 1), 7))
        for i, x in enumerate(smp):
            blk[i, :len(x)] = x
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        frames.append((i0.clone(), i1.clone(), [t] * S, cnt, blk))
    res = []
    for mode in ("frames", "batches"):
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
        if mode == "frames":
            for (i0, i1, ts, cnt, blk) in frames:
                for i in range(S):
                    trk.imu_feed_flvis(i, blk[i, :cnt[i]])
                trk.image_feed(i0, i1, ts, want_out=False, with_local_map=True)
        else:
            f = 0
            for nb in (50, 1, 1, 2, 7, 1, 3, nframes):
                nb = min(nb, nframes - f)
                if nb > 0:
                    trk.run_steps(frames[f:f + nb], with_local_map=True)
                f += nb
            assert f == nframes
        ctx.synchronize()
        rows = np.stack([trk.trajectory(i, 0, nframes) for i in range(S)])
        lms = [trk.landmarks(i) for i in (0, S - 1)]
        corr = [trk.correction(i) for i in (0, S - 1)]
        kf, ba = trk.local_map_counts()
        res.append((rows, lms, corr, trk.counters(), kf, ba))
        del trk
    (ra, la, ca, na, ka, ba_a), (rb, lb, cb, nb_, kb, ba_b) = res
    assert np.all((ra[:, 50:, 8].astype(int) & 15) == 1)
    assert ka.sum() > 4 * S and ba_a.sum() >= S                                  # keyframes, and windows that were optimised over ...
    assert np.array_equal(ka, kb) and np.array_equal(ba_a, ba_b)                 # ... the same in both runs
    assert np.array_equal(ra, rb), np.abs(ra - rb).max()
    for x, y in zip(la, lb):
        for k in ("ids", "flags", "p2d", "p2u", "p3w"):
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(ca, cb):
        assert x is not None and y is not None and x["frame_id"] == y["frame_id"]
        assert np.array_equal(x["lm_id"], y["lm_id"]) and np.array_equal(x["outlier_id"], y["outlier_id"])
        assert np.array_equal(x["pose7"], y["pose7"]) and np.array_equal(x["lm_3d"], y["lm_3d"])
    assert list(na) == list(nb_), (na, nb_)


def test_keyframe_queue_is_bounded_when_the_local_map_lags(ctx):
    """The local-map worker is launched every 8th frame only (FLVIS_BA_EVERY=8) while the batch runs flat out without readback: a
    launch that took one keyframe per stream (round 4's default) would consume a keyframe per 8 frames where the tracker makes one
    every second frame, and k_frame_end would drop keyframes at the full queue.  A launch takes at least the keyframes of the frames
    between two launches and stays while half a queue is waiting; the tracker waits in stream order: nothing is dropped, and every
    keyframe is optimised exactly once."""
    import flvis_amd
    from flvis_amd import synth
    cfg, _ = _cfgs()
    S, nframes = 16, 50 + 120
    streams = [11 + 5 * i for i in range(S)]
    old = os.environ.get("FLVIS_BA_EVERY")
    os.environ["FLVIS_BA_EVERY"] = "8"
    try:
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715)
    finally:
        if old is None:
            os.environ.pop("FLVIS_BA_EVERY", None)
        else:
            os.environ["FLVIS_BA_EVERY"] = old
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda")
    frames = []
    for f in range(nframes):        # rendered up front (kept on the device) so that the frames really run back to back
        frames.append(rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f))
    t_prev = -0.05
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i, s in enumerate(streams):
            trk.imu_feed_flvis(i, synth.imu_samples(trajs[i], s, t_prev, t))
        t_prev = t
        trk.image_feed(frames[f][0], frames[f][1], [t] * S, want_out=False, with_local_map=True)
    ctx.synchronize()
    kf, ba = trk.local_map_counts()
    assert trk.dropped_keyframes() == 0
    assert kf.min() >= 30, kf                      # a keyframe every second or third tracked frame
    # flvis_hip_synchronize drained the queues: one optimisation per keyframe once the window is full (vo_localmap.cpp:122-124)
    assert np.array_equal(ba, kf - (cfg.window_size - 1)), (kf, ba)
    del trk


@pytest.mark.gpu
def test_keyframe_hand_over_checksum_64_streams(ctx):
    """The tracker hands a keyframe to the local map with ONE agent-scope release by one thread behind a workgroup barrier (k_frame_end);
    the worker -- another workgroup, on whichever XCD the dispatcher picks -- acquires once (k_ba_worker).  With FLVIS_KF_CHECK=1 the
    producer leaves a checksum of the landmark arrays ALL of its waves wrote, and the consumer recomputes it from what it reads: 64
    streams running flat out (no read-back, the local map beside the frames), every payload checked, none torn."""
    import ctypes as C
    import flvis_amd
    from flvis_amd import synth
    cfg, _ = _cfgs()
    S, nframes = 64, 50 + 100
    old = os.environ.get("FLVIS_KF_CHECK")
    os.environ["FLVIS_KF_CHECK"] = "1"
    try:
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715)
    finally:
        if old is None:
            os.environ.pop("FLVIS_KF_CHECK", None)
        else:
            os.environ["FLVIS_KF_CHECK"] = old
    trajs = [synth.Trajectory(s) for s in range(S)]
    rnd = synth.Renderer("cuda")
    skip = cfg.skip_first_n_imgs
    frames = {f: rnd.stereo_frame(trajs, f / synth.FRAME_HZ, f) for f in range(skip, nframes)}   # (up front: the frames run back to back)
    t_prev = -0.05
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i in range(S):
            trk.imu_feed_flvis(i, synth.imu_samples(trajs[i], i, t_prev, t))
        t_prev = t
        i0, i1 = frames[max(f, skip)]   # (the skipped start-up frames are never looked at)
        trk.image_feed(i0, i1, [t] * S, want_out=False, with_local_map=True)
    ctx.synchronize()
    kf, ba = trk.local_map_counts()
    dbg = (C.c_int64 * 64)()
    ctx._check(ctx._lib.flvis_debug_counters(ctx._h, dbg), "debug_counters")
    assert trk.dropped_keyframes() == 0
    assert kf.min() >= 20, kf
    assert dbg[30] == kf.sum(), (dbg[30], kf.sum())       # every keyframe the local map took was checked ...
    assert dbg[31] == 0, "%d of %d keyframe payloads arrived torn" % (dbg[31], dbg[30])
    del trk


def _run_plain(ctx, cfg, streams, nframes, env):
    """One tracker run under the given environment knobs (read when the tracker is created): per frame the outputs that the LK feeds
    (pose, inlier counts, landmark count, landmark pixels) + the LK statistics of the run."""
    import flvis_amd
    from flvis_amd import synth
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        trk = flvis_amd.Tracker(ctx, cfg, len(streams), seed_base=0xF1715)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda")
    ctx._check(ctx._lib.flvis_debug_lk_stats(ctx._h, 1), "lk_stats")
    t_prev = -0.05
    rec = []
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        for i, s in enumerate(streams):
            trk.imu_feed_flvis(i, synth.imu_samples(trajs[i], s, t_prev, t))
        t_prev = t
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        outs = trk.image_feed(i0, i1, [t] * len(streams), with_local_map=False)
        for i in range(len(streams)):
            lm = trk.landmarks(i)
            rec.append((outs[i]["state"], outs[i]["n_landmarks"], outs[i]["dbg"].copy(), outs[i]["pose7"].copy(), lm["ids"].copy(),
                        lm["p2d"].copy(), lm["p3w"].copy()))
    dbg = (C.c_int64 * 64)()
    ctx._check(ctx._lib.flvis_debug_counters(ctx._h, dbg), "debug_counters")
    ctx._check(ctx._lib.flvis_debug_lk_stats(ctx._h, 0), "lk_stats")
    del trk
    return rec, [int(v) for v in dbg]


def test_lk_template_cache_and_pyramid_border_are_transparent(ctx):
    """The temporal LK takes its templates from the cache the previous frame's stereo LK wrote, and both stage their blocks from
    pyramids with a physical REFLECT_101 border.  Neither may change a bit: the run with both (the default) equals the run that
    computes every template and reflects every index (FLVIS_LK_TCACHE=0 FLVIS_LK_BORDER=0), frame by frame.  And the knobs do what
    they say: with the cache nearly every temporal template is taken from it, with the border almost no staging reflects indices."""
    cfg, _ = _cfgs()
    streams, nframes = [3, 140], 50 + 30
    a, ca = _run_plain(ctx, cfg, streams, nframes, {"FLVIS_LK_TCACHE": "1", "FLVIS_LK_BORDER": "1"})
    b, cb = _run_plain(ctx, cfg, streams, nframes, {"FLVIS_LK_TCACHE": "0", "FLVIS_LK_BORDER": "0"})
    assert len(a) == len(b)
    tracked = 0
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[1] == y[1], k
        for u, v in zip(x[2:], y[2:]):
            assert np.array_equal(u, v), k
        tracked += x[0] == 1
    assert tracked >= 40
    t_points = sum(ca[36 + 2 * l + 1] for l in range(6))          # temporal (point, level) pairs that iterated
    assert ca[61] >= 0.9 * t_points and ca[61] > 1000, (ca[61], t_points)
    assert cb[61] == 0
    # without the border a third of the stagings reflect indices (coarse levels: a 36 x 34 patch rarely fits an 80 x 60 image)
    assert cb[62] + cb[63] > 20 * (ca[62] + ca[63] + 1), (ca[62:64], cb[62:64])


def _patch_scene(n_patches, seed=1):
    rng = np.random.default_rng(seed)
    xs, ys, Z = rng.uniform(80, 1160, n_patches), rng.uniform(40, 330, n_patches), rng.uniform(6, 20, n_patches)
    tex = [rng.integers(0, 256, (21, 21)).astype(np.uint8) for _ in range(n_patches)]
    return xs, ys, Z, tex


def _patch_frame(xs, ys, Z, tex, keep, dx, fx=718.856, b=0.12):
    """rectified stereo pair: textured 21 x 21 patches on gray, patch i at (xs + dx, ys) on the left, shifted by its disparity on the right"""
    L = np.full((376, 1241), 110, np.uint8)
    R = L.copy()
    for i in range(len(xs)):
        if keep[i]:
            for img, x in ((L, xs[i] + dx), (R, xs[i] + dx - fx * b / Z[i])):
                x0, y0 = int(round(x)) - 10, int(round(ys[i])) - 10
                img[y0:y0 + 21, x0:x0 + 21] = tex[i]
    return L, R


@pytest.mark.parametrize("n_keep", [5, 6, 3])
def test_frontend_parity_when_the_features_run_out(ctx, n_keep):
    """A scene of 40 textured patches of which all but a few disappear after three frames: the number of LK survivors falls to
    8 .. 14, where cv::findFundamentalMat uses its LMedS registrator instead of RANSAC (fundam.cpp; with fewer than 14 points it returns
    exactly the 7 sampled points as inliers, so the reference's `< 10 F-inliers` test fails the frame), or stays above with very few
    inliers.  Every frame's state, LK / F / PnP counts, landmark count and pose are those of the oracle."""
    import flvis_amd
    import torch
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.KITTI_LIKE_YAML, "kitti_like")
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=0xF1715)
    ref = O.Tracker(ocfg, 0xF1715)
    xs, ys, Z, tex = _patch_scene(40)
    seen_lmeds = False
    prev_state = 0
    for f in range(6):
        keep = np.ones(40, bool)
        if f >= 3:
            keep[n_keep:] = False
        L, R = _patch_frame(xs, ys, Z, tex, keep, 1.0 * f)
        got = trk.image_feed(torch.from_numpy(L[None]).cuda(), torch.from_numpy(R[None]).cuda(), [0.1 * f], with_local_map=False)[0]
        want = ref.image(0.1 * f, L, R)
        assert got["state"] == want["state"] and got["n_landmarks"] == want["n_landmarks"], (f, got, want)
        if prev_state == 1:       # the frame ran LKORBTracking::tracking (otherwise the oracle's counters are those of an older frame)
            assert np.array_equal(got["dbg"], want["dbg"]), (f, got["dbg"], want["dbg"])
        assert np.array_equal(got["pose7"], want["pose7"]), (f, got["pose7"] - want["pose7"])
        prev_state = want["state"]
        seen_lmeds = seen_lmeds or (f >= 3 and 8 <= want["dbg"][0] <= 14 and want["dbg"][1] >= 7)
    if n_keep == 5:
        assert seen_lmeds          # (the 12-survivor frame of this scenario: LMedS, exactly 7 inliers)


def test_imu_staging_overflow_is_integrated_not_refused(ctx):
    """The reference integrates IMU messages as they arrive, without a limit (vo_tracking.cpp:326-371).  More than IMU_MAX = 64
    samples between two images (an IMU that leads the camera, a dropped image) must be integrated in order, not refused:
    150 samples before the first image and 90 between two later ones, compared with the oracle fed sample by sample."""
    import flvis_amd
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    rig = synth.euroc_rig()
    tr = synth.Trajectory(9)
    rnd = synth.Renderer("cuda", rig=rig)
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=0xF1715)
    ref = O.Tracker(ocfg, 0xF1715)
    # image times: 0.75 s (150 IMU samples before it), then every 50 ms, with one image "dropped" (0.45 s gap = 90 samples)
    times = [0.75, 0.80, 0.85, 1.30, 1.35, 1.40, 1.45, 1.50]
    t_prev = 0.0
    imu_want = []
    for k, t in enumerate(times):
        smp = synth.imu_samples(tr, 9, t_prev, t)
        if k in (0, 3):
            assert len(smp) > 64
        trk.imu_feed_flvis(0, smp)
        for r in smp:
            imu_want.append(np.concatenate([[r[0]], ref.imu(r[0], r[1:4], r[4:7])]))
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, k)
        got = trk.image_feed(i0, i1, [t], with_local_map=False)[0]
        want = ref.image(t, i0[0].cpu().numpy(), i1[0].cpu().numpy())
        assert got["state"] == want["state"] and got["n_landmarks"] == want["n_landmarks"], k
        assert np.abs(got["pose7"] - want["pose7"]).max() < 1e-6, (k, got["pose7"] - want["pose7"])
    assert want["state"] == 1
    rows, dropped = trk.imu_states(0)                    # all 300 samples, through the flushes, in order
    assert dropped == 0 and np.array_equal(rows, np.array(imu_want))


def test_imu_feed_out_call_for_call(ctx):
    """flvis_imu_feed_out = F2FTracking::imu_feed(time, acc, gyro, q_w_i&, pos_w_i&, vel_w_i&) (f2f_tracking.cpp:46-57) sample by
    sample in the SENSOR frame (EuRoC remap of imu_callback, vo_tracking.cpp:341-349): identity / zeros during the attitude
    initialisation, the propagated state afterwards, re-anchored by the vision corrections of the images in between --
    bit-identical to the oracle's outputs, and the batched getter returns the same rows afterwards; a ring that wrapped
    reports the rows it lost."""
    import flvis_amd
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    assert cfg.imu_type == 1
    tr = synth.Trajectory(9)
    rnd = synth.Renderer("cuda", rig=synth.euroc_rig())
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=0xF1715)
    ref = O.Tracker(ocfg, 0xF1715)
    t_prev, want_rows, n_moving = 0.0, [], 0
    for k in range(14):
        t = 0.30 + k / synth.FRAME_HZ
        for r in synth.imu_samples(tr, 9, t_prev, t):
            acc_s = [-r[3], r[2], -r[1]]               # inverse of acc = (-z, y, -x)
            gyro_s = [r[6], -r[5], r[4]]               # inverse of gyro = (z, -y, x)
            q, p, v = trk.imu_feed_out(0, r[0], acc_s, gyro_s)
            w = ref.imu(r[0], r[1:4], r[4:7])
            assert np.array_equal(np.concatenate([q, p, v]), w), (k, r[0])
            want_rows.append(np.concatenate([[r[0]], w]))
            n_moving += int(np.any(w[4:] != 0))
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, k)
        got = trk.image_feed(i0, i1, [t], with_local_map=False)[0]
        want = ref.image(t, i0[0].cpu().numpy(), i1[0].cpu().numpy())
        assert got["state"] == want["state"] and np.array_equal(got["pose7"], want["pose7"]), k
    assert want["state"] == 1 and n_moving > 50        # propagation (not only the initialisation) was compared
    rows, dropped = trk.imu_states(0, cap=100)           # the batched form hands out the same rows, cap at a time
    assert dropped == 0 and np.array_equal(rows, np.array(want_rows)[:100])
    rows2, _ = trk.imu_states(0)
    assert np.array_equal(rows2, np.array(want_rows)[100:])
    smp = synth.imu_samples(tr, 9, t_prev, t_prev + 3.0)  # 600 samples without a fetch: the 512-row ring wraps
    trk.imu_feed_flvis(0, smp)
    rows3, dropped = trk.imu_states(0)
    assert dropped == len(smp) - 512 and len(rows3) == 512 and np.array_equal(rows3[:, 0], smp[-512:, 0])


def test_trajectory_recorder(ctx):
    """flvis_write_trajectory (the recorder of vo_repub_rec.cpp:74-124): both file formats agree with the device
    trajectory, only TRACKING frames are written, the throttle rule holds."""
    import flvis_amd
    from flvis_amd import synth, traj_io
    cfg, _ = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    rig = synth.euroc_rig()
    nframes = 30
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=5, traj_capacity=nframes)
    tr = synth.Trajectory(9)
    rnd = synth.Renderer("cuda", rig=rig)
    t_prev = -0.05
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        trk.imu_feed_flvis(0, synth.imu_samples(tr, 9, t_prev, t))
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, f)
        trk.image_feed(i0, i1, [t], with_local_map=False)
    rows = trk.trajectory(0, 0, nframes)
    tracked = [i for i in range(nframes) if (int(rows[i, 8]) & 15) == 1]
    assert len(tracked) >= 15
    p0 = os.path.join(tempfile.gettempdir(), "flvis_rec_tum.txt")
    p1 = os.path.join(tempfile.gettempdir(), "flvis_rec_kitti.txt")
    assert trk.write_trajectory(0, 0, nframes, p0, 0) == len(tracked)
    assert trk.write_trajectory(0, 0, nframes, p1, 1) == len(tracked)
    ts, pos, quat = traj_io.read_stamped(p0)
    Rk, tk = traj_io.read_kitti(p1)
    for k, i in enumerate(tracked):
        R = traj_io.quat_to_rot(rows[i, 7], rows[i, 4], rows[i, 5], rows[i, 6])      # T_c_w rotation
        centre = -R.T @ rows[i, 1:4]
        assert abs(ts[k] - rows[i, 0]) < 1e-8
        assert np.allclose(pos[k], centre, rtol=1e-5, atol=1e-6) and np.allclose(tk[k], centre, rtol=1e-5, atol=1e-6)
        assert np.allclose(Rk[k], R.T, atol=1e-5)
        assert np.allclose(traj_io.quat_to_rot(*quat[k]), R.T, atol=1e-5)           # written orientation is q_w_c
    n_thr = trk.write_trajectory(0, 0, nframes, p0, 0, 0.1)
    assert n_thr == len(traj_io.throttle(rows[tracked, 0], 0.1))
    # the estimate follows the synthetic ground truth (camera centres, rigid alignment)
    gt = np.array([-(tr.T_c_w(rows[i, 0], rig)[0]).T @ tr.T_c_w(rows[i, 0], rig)[1] for i in tracked])
    assert traj_io.ate_rmse(pos, gt) < 0.02


def _perturbed_pose(pose7, dt, drot):
    """pose7 (t, q xyzw) composed with a small translation / rotation-vector perturbation."""
    t = np.asarray(pose7[:3]) + np.asarray(dt)
    q = np.asarray(pose7[3:7])
    th = np.linalg.norm(drot)
    dq = np.concatenate([np.sin(th / 2) * np.asarray(drot) / th, [np.cos(th / 2)]])
    x1, y1, z1, w1 = dq
    x2, y2, z2, w2 = q
    qq = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                   w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    return np.concatenate([t, qq / np.linalg.norm(qq)])


def test_local_map_feedback_parity(ctx):
    """F2FTracking::correction_feed + STEP1 of the Tracking case (f2f_tracking.cpp:40-44,189-219; dead in the
    reference's v2, opt-in here).  The same synthetic corrections are fed to the HIP tracker and to the CPU restatement:
    one naming a recorded keyframe pose, one naming an unknown frame id (the reference then falls back to the OLDEST
    record), each with corrected landmark positions and outlier ids.  Afterwards the two runs must keep agreeing frame
    by frame (pose_records, poses, landmark sets), and must both differ from a run without the feedback."""
    import flvis_amd
    from flvis_amd import synth
    cfg, ocfg = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    rig = synth.euroc_rig()
    nframes, sid = 34, 9
    tr = synth.Trajectory(sid)
    rnd = synth.Renderer("cuda", rig=rig)
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=0xF1715)
    ref = O.Tracker(ocfg, 0xF1715)
    ref_nofb = O.Tracker(ocfg, 0xF1715)
    t_prev = -0.05
    tracked = 0
    lock = True
    fed = []
    checked_after_feed = 0
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        smp = synth.imu_samples(tr, sid, t_prev, t)
        trk.imu_feed_flvis(0, smp)
        for r in smp:
            ref.imu(r[0], r[1:4], r[4:7])
            ref_nofb.imu(r[0], r[1:4], r[4:7])
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, f)
        got = trk.image_feed(i0, i1, [t], with_local_map=False)[0]
        h0, h1 = i0.cpu().numpy()[0], i1.cpu().numpy()[0]
        want = ref.image(t, h0, h1)
        base = ref_nofb.image(t, h0, h1)
        where = "frame %d" % f
        assert got["state"] == want["state"] and got["new_keyframe"] == want["new_keyframe"], where
        gl, wl = trk.landmarks(0), ref.landmarks()
        # lockstep: the corrected run is bit-identical on both sides, frame by frame
        assert np.array_equal(got["pose7"], want["pose7"]), (where, got["pose7"] - want["pose7"])
        assert got["n_landmarks"] == want["n_landmarks"] and np.array_equal(gl["ids"], wl["ids"]) and np.array_equal(gl["flags"], wl["flags"]), where
        grec, wrec = trk.pose_records(0), ref.pose_records()
        assert len(grec) == len(wrec) and np.array_equal(grec, wrec), (where, np.abs(grec - wrec).max())
        if fed and fed[-1]["frame"] == f - 1:
            # the frame right after a correction: the correction must have reached both sides identically ...
            checked_after_feed += 1
            assert np.array_equal(gl["p3w"], wl["p3w"]), where
            # ... and must have had an effect: the pose of the next frame comes from PnP on the corrected landmarks
            # (useExtrinsicGuess=false, so last_frame->T_c_w itself does not enter), 1 cm on a third of them moves it ~0.5 mm
            assert np.abs(want["pose7"] - base["pose7"]).max() > 1e-4, where
            # re-anchoring: the named record now carries the corrected pose
            k = fed[-1]["rec_index"]
            assert np.allclose(wrec[k, 1:], fed[-1]["pose7"], atol=1e-9), where
        if want["state"] == 1:
            tracked += 1
            if tracked in (3, 7):
                ids = wl["ids"]
                sel = np.arange(0, len(ids), 3)
                lm_3d = wl["p3w"][sel] + 0.01 * np.sin(np.arange(len(sel) * 3)).reshape(-1, 3)
                outl = ids[1::7]
                recs = ref.pose_records()
                if tracked == 3:
                    fid = int(recs[-2, 0])               # a recorded frame: the one before the current
                    k = len(recs) - 2
                else:
                    fid = 100000                          # unknown frame id: falls back to the oldest record
                    k = 0
                pose = _perturbed_pose(recs[k, 1:], [0.012, -0.02, 0.015], [0.004, -0.003, 0.005])
                trk.correction_feed(0, fid, pose, ids[sel], lm_3d, outl)
                ref.correction_feed(fid, pose, ids[sel], lm_3d, outl)
                fed.append(dict(frame=f, rec_index=k, pose7=pose, locked=lock))
    assert len(fed) == 2 and checked_after_feed == 2
    assert fed[0]["locked"], "lockstep was lost before the first correction: the exact comparison did not happen"
    assert tracked >= 20


def test_local_map_feedback_closed_loop(ctx):
    """The loop the paper title describes: every CorrectionInf the local map produces is fed back into the tracker
    (flvis_get_correction -> flvis_correction_feed).  Tracking must survive it and stay on the synthetic ground truth."""
    import flvis_amd
    from flvis_amd import synth, traj_io
    cfg, _ = _cfgs_yaml(synth.EUROC_LIKE_YAML, "euroc_like")
    rig = synth.euroc_rig()
    nframes, sid = 60, 9
    tr = synth.Trajectory(sid)
    rnd = synth.Renderer("cuda", rig=rig)
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=3, traj_capacity=nframes)
    t_prev = -0.05
    n_fed = 0
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        trk.imu_feed_flvis(0, synth.imu_samples(tr, sid, t_prev, t))
        t_prev = t
        i0, i1 = rnd.stereo_frame([tr], t, f)
        out = trk.image_feed(i0, i1, [t], with_local_map=True)[0]
        if out["new_keyframe"]:
            c = trk.correction(0)
            if c is not None:
                trk.correction_feed(0, c["frame_id"], c["pose7"], c["lm_id"], c["lm_3d"], c["outlier_id"])
                n_fed += 1
    rows = trk.trajectory(0, 0, nframes)
    tracked = [i for i in range(nframes) if (int(rows[i, 8]) & 15) == 1]
    assert n_fed >= 3 and len(tracked) >= 45
    est = np.array([-(traj_io.quat_to_rot(rows[i, 7], rows[i, 4], rows[i, 5], rows[i, 6])).T @ rows[i, 1:4] for i in tracked])
    gt = np.array([-(tr.T_c_w(rows[i, 0], rig)[0]).T @ tr.T_c_w(rows[i, 0], rig)[1] for i in tracked])
    assert traj_io.ate_rmse(est, gt) < 0.05


def _quat_wxyz(R):
    import _geom as G
    p7 = G.pose7(R, np.zeros(3))
    return np.array([p7[6], p7[3], p7[4], p7[5]])


def test_local_map_parity_with_imu_factor(ctx):
    """The window BA with the optional IMU rotation edges (flvis_set_imu_factor / flvis_ba_push_keyframe_imu) against the
    oracle's BAGraph with the same edges: every keyframe comes with the true relative body rotation + 1 mrad noise."""
    import flvis_amd
    import _geom as G
    cfg, _ = _cfgs()
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    T_i_c = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    Rcb = T_i_c[:3, :3].T                                        # camera <- body
    sigma_g = 0.004
    # the keyframes of the first sequence without the factor (a context holds one tracker at a time: this one goes first)
    plain = flvis_amd.Tracker(ctx, cfg, 1, seed_base=1)
    seq0 = B.make_sequence(21, n_kf=14, n_lm=260, outlier_frac=0.03)
    base_out = [plain.ba_push_keyframe(0, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"]) for kf in seq0["kfs"]]
    del plain
    trk = flvis_amd.Tracker(ctx, cfg, 2, seed_base=1)
    for stream, seed in ((0, 21), (1, 22)):
        seq = B.make_sequence(seed, n_kf=14, n_lm=260, outlier_frac=0.03)
        rng = np.random.default_rng(seed)
        ref = O.LocalMap(cfg.window_size, K4)
        ref.set_imu_factor(True, sigma_g, _quat_wxyz(Rcb))
        trk.set_imu_factor(True, sigma_g)
        produced, moved = 0, 0.0
        for k, kf in enumerate(seq["kfs"]):
            dq, dt = None, 0.0
            if k > 0:
                Ra, Rb = seq["gt"][k - 1][0], seq["gt"][k][0]
                dR = (Ra.T @ Rcb).T @ (Rb.T @ Rcb) @ G.rodrigues(rng.normal(0, 1e-3, 3))
                dq, dt = _quat_wxyz(dR), 0.1 + 0.02 * (k % 3)
                ref.next_imu(dq, dt)
            want = ref.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
            got = trk.ba_push_keyframe(stream, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"], imu_dq=dq, imu_dt=dt)
            base = base_out[k] if stream == 0 else None
            assert (want is None) == (got is None), k
            if want is None:
                continue
            produced += 1
            assert got["frame_id"] == want["frame_id"]
            assert np.array_equal(got["lm_id"], want["lm_id"]), k
            assert np.array_equal(got["outlier_id"], want["outlier_id"]), k
            # fp64 LM chain with a different summation order: 1e-6 on pose and landmarks, as for the reprojection-only window
            assert np.allclose(got["pose7"], want["pose7"], atol=1e-6, rtol=0), (k, got["pose7"] - want["pose7"])
            assert np.allclose(got["lm_3d"], want["lm_3d"], atol=1e-6, rtol=0), (k, np.abs(got["lm_3d"] - want["lm_3d"]).max())
            if base is not None:
                moved = max(moved, np.abs(got["pose7"] - base["pose7"]).max())
        assert produced == len(seq["kfs"]) - cfg.window_size + 1
        if stream == 0:
            assert moved > 1e-4          # the edges do change the solution (the parity above is not vacuous)


def test_local_map_parity_with_the_full_imu_factor(ctx):
    """The window BA with rotation AND position rows (flvis_set_imu_factor + flvis_set_imu_factor_accel, flvis_ba_push_keyframe_imu_pos)
    against the oracle's BAGraph with the same 6-row edges: every keyframe comes with the true relative body rotation and the true
    preintegrated displacement (+ noise), for an extrinsic with a lever arm (the D435i's T_imu_cam0)."""
    import flvis_amd
    import _geom as G
    cfg, _ = _cfgs()
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    T_i_c = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    Rcb = T_i_c[:3, :3].T                                        # camera <- body
    tcb = -Rcb @ T_i_c[:3, 3]
    gw = np.array([0.0, 0.0, -9.81])
    sigma_g, sigma_a = 0.004, 0.08
    trk = flvis_amd.Tracker(ctx, cfg, 2, seed_base=1)
    trk.set_imu_factor(True, sigma_g)
    trk.set_imu_factor_accel(sigma_a)
    rot_only = {}
    for stream, seed in ((0, 21), (1, 22)):
        seq = B.make_sequence(seed, n_kf=14, n_lm=260, outlier_frac=0.03)
        rng = np.random.default_rng(seed)
        ref = O.LocalMap(cfg.window_size, K4)
        ref.set_imu_factor(True, sigma_g, _quat_wxyz(Rcb))
        ref.set_imu_factor_pos(sigma_a, tcb)

        def body(k):
            R, t = seq["gt"][k]
            return R.T @ Rcb, R.T @ (tcb - t)
        produced = 0
        for k, kf in enumerate(seq["kfs"]):
            dq, dt, dp, va = None, 0.0, None, None
            if k > 0:
                Rwa, pa = body(k - 1)
                Rwb, pb = body(k)
                dt = 0.1 + 0.02 * (k % 3)
                dq = _quat_wxyz(Rwa.T @ Rwb @ G.rodrigues(rng.normal(0, 1e-3, 3)))
                va = (pb - pa) / dt + rng.normal(0, 0.05, 3)
                dp = Rwa.T @ (pb - pa - va * dt + 0.5 * gw * dt * dt) + rng.normal(0, 2e-3, 3)
                ref.next_imu(dq, dt)
                ref.next_imu_pos(dp, va)
            want = ref.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
            got = trk.ba_push_keyframe(stream, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"], imu_dq=dq, imu_dt=dt,
                                       imu_dp=dp, imu_va=va)
            assert (want is None) == (got is None), k
            if want is None:
                continue
            produced += 1
            assert got["frame_id"] == want["frame_id"] and np.array_equal(got["lm_id"], want["lm_id"]), k
            assert np.array_equal(got["outlier_id"], want["outlier_id"]), k
            assert np.allclose(got["pose7"], want["pose7"], atol=1e-6, rtol=0), (k, got["pose7"] - want["pose7"])
            assert np.allclose(got["lm_3d"], want["lm_3d"], atol=1e-6, rtol=0), (k, np.abs(got["lm_3d"] - want["lm_3d"]).max())
            rot_only[(stream, k)] = got["pose7"]
        assert produced == len(seq["kfs"]) - cfg.window_size + 1
    # the position rows do change the solution: the same keyframes with the rotation rows alone end elsewhere
    trk.set_imu_factor_accel(0.0)
    del trk
    trk2 = flvis_amd.Tracker(ctx, cfg, 1, seed_base=1)
    trk2.set_imu_factor(True, sigma_g)
    seq = B.make_sequence(21, n_kf=14, n_lm=260, outlier_frac=0.03)
    rng = np.random.default_rng(21)
    moved = 0.0
    for k, kf in enumerate(seq["kfs"]):
        dq, dt = None, 0.0
        if k > 0:
            R0, t0 = seq["gt"][k - 1]
            R1, t1 = seq["gt"][k]
            dt = 0.1 + 0.02 * (k % 3)
            dq = _quat_wxyz((R0.T @ Rcb).T @ (R1.T @ Rcb) @ G.rodrigues(rng.normal(0, 1e-3, 3)))
            rng.normal(0, 0.05, 3), rng.normal(0, 2e-3, 3)        # (the draws of the run above)
        got = trk2.ba_push_keyframe(0, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"], imu_dq=dq, imu_dt=dt)
        if got is not None:
            moved = max(moved, np.abs(got["pose7"] - rot_only[(0, k)]).max())
    assert moved > 1e-5


@pytest.mark.parametrize("pos_rows", [False, True])
def test_local_map_imu_factor_at_window_size_16(ctx, pos_rows):
    """The largest window the solver holds (BA_WMAX = 16) with the IMU factor on: 15 edges between consecutive keyframes, i.e. 15 x 36 =
    540 entries of off-diagonal 6 x 6 blocks in the reduced system -- more than the workgroup has threads.  Every one of them must
    arrive (the mapping strides), for the rotation rows alone (9 live entries per block) and with the position rows (36)."""
    import flvis_amd
    import _geom as G
    cfg, _ = _cfgs()
    cfg.window_size = 16
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    T_i_c = np.array(list(cfg.T_imu_cam0)).reshape(4, 4)
    Rcb = T_i_c[:3, :3].T
    tcb = -Rcb @ T_i_c[:3, 3]
    gw = np.array([0.0, 0.0, -9.81])
    sigma_g, sigma_a = 0.004, 0.08
    trk = flvis_amd.Tracker(ctx, cfg, 1, seed_base=1)
    trk.set_imu_factor(True, sigma_g)
    if pos_rows:
        trk.set_imu_factor_accel(sigma_a)
    seq = B.make_sequence(31, n_kf=21, n_lm=220, outlier_frac=0.03)
    rng = np.random.default_rng(31)
    ref = O.LocalMap(16, K4)
    ref.set_imu_factor(True, sigma_g, _quat_wxyz(Rcb))
    if pos_rows:
        ref.set_imu_factor_pos(sigma_a, tcb)

    def body(k):
        R, t = seq["gt"][k]
        return R.T @ Rcb, R.T @ (tcb - t)
    produced = 0
    for k, kf in enumerate(seq["kfs"]):
        dq, dt, dp, va = None, 0.0, None, None
        if k > 0:
            Rwa, pa = body(k - 1)
            Rwb, pb = body(k)
            dt = 0.1 + 0.02 * (k % 3)
            dq = _quat_wxyz(Rwa.T @ Rwb @ G.rodrigues(rng.normal(0, 1e-3, 3)))
            ref.next_imu(dq, dt)
            if pos_rows:
                va = (pb - pa) / dt + rng.normal(0, 0.05, 3)
                dp = Rwa.T @ (pb - pa - va * dt + 0.5 * gw * dt * dt) + rng.normal(0, 2e-3, 3)
                ref.next_imu_pos(dp, va)
        want = ref.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
        got = trk.ba_push_keyframe(0, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"], imu_dq=dq, imu_dt=dt, imu_dp=dp, imu_va=va)
        assert (want is None) == (got is None), k
        if want is None:
            continue
        produced += 1
        assert got["frame_id"] == want["frame_id"] and np.array_equal(got["lm_id"], want["lm_id"]), k
        assert np.array_equal(got["outlier_id"], want["outlier_id"]), k
        assert np.allclose(got["pose7"], want["pose7"], atol=1e-6, rtol=0), (k, got["pose7"] - want["pose7"])
        assert np.allclose(got["lm_3d"], want["lm_3d"], atol=1e-6, rtol=0), (k, np.abs(got["lm_3d"] - want["lm_3d"]).max())
    assert produced == len(seq["kfs"]) - 16 + 1
    if pos_rows:
        trk.set_imu_factor_accel(0.0)


@pytest.mark.parametrize("balance", ["0", "1"])
def test_local_map_parity(ctx, monkeypatch, balance):
    """The window optimiser against the oracle, keyframe by keyframe.  balance = 1: the Schur accumulate's lanes dealt to the pose pairs in
    proportion to the landmarks they share (FLVIS_BA_BALANCE=1, opt-in; profiles/r06_ba_phases.md) -- another order of the same sums, the
    same tolerance."""
    import flvis_amd
    cfg, _ = _cfgs()
    monkeypatch.setenv("FLVIS_BA_BALANCE", balance)
    trk = flvis_amd.Tracker(ctx, cfg, 2, seed_base=1)
    K4 = np.array([cfg.P0[0], cfg.P0[5], cfg.P0[2], cfg.P0[6]])
    for stream, seed in ((0, 11), (1, 12)):
        seq = B.make_sequence(seed, n_kf=14, n_lm=260, outlier_frac=0.03)
        ref = O.LocalMap(cfg.window_size, K4)
        produced = 0
        for k, kf in enumerate(seq["kfs"]):
            want = ref.push(kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
            got = trk.ba_push_keyframe(stream, kf["frame_id"], kf["pose7"], kf["lm_id"], kf["lm_2d"], kf["lm_3d"])
            assert (want is None) == (got is None), k
            if want is None:
                continue
            produced += 1
            assert got["frame_id"] == want["frame_id"]
            assert np.array_equal(got["lm_id"], want["lm_id"]), k                      # index work: exact
            assert np.array_equal(got["outlier_id"], want["outlier_id"]), k            # descending edge-id order
            # fp64 LM chain (20 iterations, different summation order): tolerance 1e-6 m on pose and landmarks
            assert np.allclose(got["pose7"], want["pose7"], atol=1e-6, rtol=0), (k, got["pose7"] - want["pose7"])
            assert np.allclose(got["lm_3d"], want["lm_3d"], atol=1e-6, rtol=0), (
                k, np.abs(got["lm_3d"] - want["lm_3d"]).max())
        assert produced == len(seq["kfs"]) - cfg.window_size + 1


def _mode_frames(S, nframes, streams):
    from flvis_amd import synth
    trajs = [synth.Trajectory(s) for s in streams]
    rnd = synth.Renderer("cuda")
    frames, t_prev = [], -0.05
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        smp = [synth.imu_samples(trajs[i], s, t_prev, t) for i, s in enumerate(streams)]
        t_prev = t
        cnt = np.array([len(x) for x in smp], np.int32)
        blk = np.zeros((S, max(max(len(x) for x in smp), 1), 7))
        for i, x in enumerate(smp):
            blk[i, :len(x)] = x
        i0, i1 = rnd.stereo_frame(trajs, t, f)
        frames.append((i0.clone(), i1.clone(), [t] * S, cnt, blk))
    return frames


def _mode_result(trk, ctx, S, nframes):
    ctx.synchronize()
    rows = np.stack([trk.trajectory(i, 0, nframes) for i in range(S)])
    lms = [trk.landmarks(i) for i in (0, S - 1)]
    corr = [trk.correction(i) for i in (0, S - 1)]
    kf, ba = trk.local_map_counts()
    return rows, lms, corr, list(trk.counters()), kf, ba


def _assert_same_run(a, b, what):
    (ra, la, ca, na, ka, ba_a), (rb, lb, cb, nb_, kb, ba_b) = a, b
    assert np.array_equal(ka, kb) and np.array_equal(ba_a, ba_b), what
    assert np.array_equal(ra, rb), (what, np.abs(ra - rb).max())
    for x, y in zip(la, lb):
        for k in ("ids", "flags", "p2d", "p2u", "p3w"):
            assert np.array_equal(x[k], y[k]), (what, k)
    for x, y in zip(ca, cb):
        assert (x is None) == (y is None), what
        if x is not None:
            assert x["frame_id"] == y["frame_id"] and np.array_equal(x["pose7"], y["pose7"]) and np.array_equal(x["lm_3d"], y["lm_3d"]), what
    assert na == nb_, (what, na, nb_)


def test_stream_join_forms_leave_the_same_results(ctx, monkeypatch):
    """Round 6: the joins between a lane's streams are device words (k_store_flag / k_wait_flag) and, where the chain's own kernels can
    carry them, folded into those kernels (KJoin) -- instead of hipEventRecord / hipStreamWaitEvent.  FLVIS_JOIN=event (rounds 1-5),
    flag joins without folding and the default must leave the same trajectories, landmarks, CorrectionInf and counters, bit for bit --
    frame by frame (per-frame callers: the explicit frame-start join) and in batches (flvis_run_steps: the folded one, the deferred
    local-map launch behind k_ransac_pnp)."""
    import flvis_amd
    cfg, _ = _cfgs()
    S, nframes = 8, 50 + 36
    frames = _mode_frames(S, nframes, [3 + 7 * i for i in range(S)])
    res = {}
    for name, env in (("event", {"FLVIS_JOIN": "event"}), ("flag", {"FLVIS_JOIN_FOLD": "0"}), ("default", {})):
        for k in ("FLVIS_JOIN", "FLVIS_JOIN_FOLD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for feed in ("frames", "batches"):
            trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
            if feed == "frames":
                for (i0, i1, ts, cnt, blk) in frames:
                    for i in range(S):
                        trk.imu_feed_flvis(i, blk[i, :cnt[i]])
                    trk.image_feed(i0, i1, ts, want_out=False, with_local_map=True)
            else:
                f = 0
                for nb in (50, 1, 5, 2, nframes):
                    nb = min(nb, nframes - f)
                    if nb > 0:
                        trk.run_steps(frames[f:f + nb], with_local_map=True)
                    f += nb
            res[(name, feed)] = _mode_result(trk, ctx, S, nframes)
            del trk
    ref = res[("event", "frames")]
    assert np.all((ref[0][:, 50:, 8].astype(int) & 15) == 1) and ref[4].sum() > 2 * S and ref[5].sum() >= 1
    for key, r in res.items():
        _assert_same_run(ref, r, key)


def test_templates_ahead_leave_the_same_results(ctx, monkeypatch):
    """Round 6: the stereo matcher's templates of the landmarks the temporal tracker has followed into the frame are computed AHEAD, by
    k_lk_templates_ahead on a stream of its own beside the frame's geometry kernels, and the stereo launch takes them from the cache
    (FLVIS_TPL_AHEAD=1, an opt-in knob: measured, not faster -- profiles/r06_templates_ahead.md).  With them, with the kernel started behind
    k_track_collect instead of behind the F-RANSAC (FLVIS_TPL_START=0), with event joins, and without them (the default: rounds 4-5) a
    run must leave the same trajectories, landmarks, CorrectionInf and counters, bit for bit -- frame by frame and in batches."""
    import flvis_amd
    cfg, _ = _cfgs()
    S, nframes = 8, 50 + 36
    frames = _mode_frames(S, nframes, [5 + 3 * i for i in range(S)])
    res = {}
    knobs = ("FLVIS_TPL_AHEAD", "FLVIS_TPL_START", "FLVIS_JOIN")
    for name, env in (("off", {}), ("ahead", {"FLVIS_TPL_AHEAD": "1"}), ("start0", {"FLVIS_TPL_AHEAD": "1", "FLVIS_TPL_START": "0"}),
                      ("event", {"FLVIS_TPL_AHEAD": "1", "FLVIS_JOIN": "event"})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for feed in ("frames", "batches"):
            trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
            if feed == "frames":
                for (i0, i1, ts, cnt, blk) in frames:
                    for i in range(S):
                        trk.imu_feed_flvis(i, blk[i, :cnt[i]])
                    trk.image_feed(i0, i1, ts, want_out=False, with_local_map=True)
            else:
                f = 0
                for nb in (50, 1, 5, 2, nframes):
                    nb = min(nb, nframes - f)
                    if nb > 0:
                        trk.run_steps(frames[f:f + nb], with_local_map=True)
                    f += nb
            res[(name, feed)] = _mode_result(trk, ctx, S, nframes)
            del trk
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
    ref = res[("off", "frames")]
    assert np.all((ref[0][:, 50:, 8].astype(int) & 15) == 1) and ref[4].sum() > 2 * S and ref[5].sum() >= 1
    for key, r in res.items():
        _assert_same_run(ref, r, key)


def test_chain_merge_forms_leave_the_same_results(ctx, monkeypatch):
    """Round 6: launches of the frame's chain folded into their neighbours (FLVIS_CHAIN_MERGE, bits: 1 = k_add_new + k_depth_seeds as one
    launch, 2 = k_track_collect as the prologue of k_ransac_f, by sixteen waves instead of one; default 3).  Every combination must leave
    the same trajectories, landmarks, CorrectionInf and counters as rounds 1-5's launches (0), bit for bit -- frame by frame and in batches."""
    import flvis_amd
    cfg, _ = _cfgs()
    S, nframes = 8, 50 + 36
    frames = _mode_frames(S, nframes, [2 + 5 * i for i in range(S)])
    res = {}
    for merge in ("0", "1", "2", "3", None):
        monkeypatch.delenv("FLVIS_CHAIN_MERGE", raising=False)
        if merge is not None:
            monkeypatch.setenv("FLVIS_CHAIN_MERGE", merge)
        for feed in ("frames", "batches"):
            trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
            if feed == "frames":
                for (i0, i1, ts, cnt, blk) in frames:
                    for i in range(S):
                        trk.imu_feed_flvis(i, blk[i, :cnt[i]])
                    trk.image_feed(i0, i1, ts, want_out=False, with_local_map=True)
            else:
                f = 0
                for nb in (50, 1, 5, 2, nframes):
                    nb = min(nb, nframes - f)
                    if nb > 0:
                        trk.run_steps(frames[f:f + nb], with_local_map=True)
                    f += nb
            res[(merge, feed)] = _mode_result(trk, ctx, S, nframes)
            del trk
    monkeypatch.delenv("FLVIS_CHAIN_MERGE", raising=False)
    ref = res[("0", "frames")]
    assert np.all((ref[0][:, 50:, 8].astype(int) & 15) == 1) and ref[4].sum() > 2 * S and ref[5].sum() >= 1
    for key, r in res.items():
        _assert_same_run(ref, r, key)


def test_host_feed_upload_forms_leave_the_same_results(ctx, monkeypatch):
    """flvis_image_feed_host, mode 2 (round 6: nothing but copies on the copy stream -- a sequence block behind the images, k_wait_flag on the
    ingesting stream, host-side slot gating over three staging slots) against mode 1 (FLVIS_H2D_MODE=1: events on the copy stream, two
    slots; rounds 4-5) and against the resident entry: 40 frames handed over without readback, buffers rewritten as soon as the contract
    allows, must leave the same trajectories, landmarks and counters, bit for bit."""
    import flvis_amd
    cfg, _ = _cfgs()
    S, nframes = 8, 50 + 40
    frames = _mode_frames(S, nframes, [11 + 3 * i for i in range(S)])
    res = {}
    for mode in ("resident", "2", "1"):
        monkeypatch.delenv("FLVIS_H2D_MODE", raising=False)
        if mode == "1":
            monkeypatch.setenv("FLVIS_H2D_MODE", "1")
        trk = flvis_amd.Tracker(ctx, cfg, S, seed_base=0xF1715, traj_capacity=nframes)
        feeder = None if mode == "resident" else _HostFeeder(S, channels=1, hold=1, layout="block", pinned=True)
        for (i0, i1, ts, cnt, blk) in frames:
            for i in range(S):
                trk.imu_feed_flvis(i, blk[i, :cnt[i]])
            if feeder is None:
                trk.image_feed(i0, i1, ts, want_out=False, with_local_map=True)
            else:
                feeder.feed(trk, i0.cpu().numpy(), i1.cpu().numpy(), ts, False, want_out=False, with_local_map=True)
        res[mode] = _mode_result(trk, ctx, S, nframes)
        del trk
    for mode in ("2", "1"):
        _assert_same_run(res["resident"], res[mode], "host-image mode " + mode)
