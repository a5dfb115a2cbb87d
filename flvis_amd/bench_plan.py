"""Frame schedule of bench.py as pure functions (no torch, no GPU), so that it can be tested on CPU.

The reference drops the first `skip_first_n_imgs` frames of a stream before any processing
(src/frontend/f2f_tracking.cpp:120-124,136-140; 50 for the D435 modes, 0 for EuRoC/KITTI), needs >30 IMU samples for the
attitude initialisation and one `init_frame` before a stream is in the Tracking state.  A timed region that starts before
that measures skip-path frames (no vision work).  The schedule therefore always runs an UNTIMED pre-roll of
skip + SETTLE frames (+ up to EXTRA_SETTLE_MAX more while some stream is not yet tracking) before the caller's --warmup.

The local map does not optimise until a stream's window holds `window_size` keyframes (vo_localmap.cpp:211-214): the first 40 frames
emit a keyframe every fifth frame (f2f_tracking.cpp:339-354), so the first optimisation of a stream comes ~35-40 tracked frames
after its init_frame.  A timed region before that contains no optimiser work (round 2 timed exactly that with the driver's
--warmup 5).  With the local map on, the pre-roll therefore also continues until EVERY stream has run at least one optimisation
(`steady_state`), after which every keyframe triggers one: the region is BA-steady-state for any --warmup / --steps.
"""

SETTLE = 12            # tracked frames fed after the skipped ones before the state of every stream is checked
EXTRA_SETTLE_MAX = 90  # additional pre-roll frames while a stream is still not tracking / has not yet run its first optimisation
EPILOGUE = 20          # untimed frames after the timed region: every stage bracketed by HIP events (costs ~10 %)


def frame_schedule(steps, warmup, skip, settle=SETTLE, epilogue=EPILOGUE, extra_settle=0):
    """Half-open frame ranges of one bench run.  Frame f of every stream has stamp f / FRAME_HZ.

    preroll  [0, p)            untimed, p = skip + settle + extra_settle: the skipped start-up frames, init_frame, first
                               tracked frames; ends with every stream in the Tracking state (checked by the caller)
    warmup   [p, p + W)        untimed, the caller's --warmup
    timed    [p + W, p + W + K)  the K timed steps -- always steady-state tracking frames
    epilogue [.., + epilogue)  untimed per-stage event timing
    """
    steps, warmup, skip = int(steps), int(warmup), int(skip)
    if steps < 1:
        raise ValueError("--steps must be >= 1")
    if warmup < 0 or skip < 0 or settle < 1 or epilogue < 0 or extra_settle < 0:
        raise ValueError("bad schedule arguments")
    p = skip + settle + extra_settle
    w0, t0 = p, p + warmup
    t1 = t0 + steps
    return {"skip": skip, "preroll": (0, p), "warmup": (w0, t0), "timed": (t0, t1), "epilogue": (t1, t1 + epilogue),
            "n_frames": t1 + epilogue}


def steady_state(n_tracking, n_streams, ba_runs_per_stream, local_map):
    """The pre-roll may end: every stream tracks and (with the local map on) every stream's window has optimised at least once."""
    if n_tracking != n_streams:
        return False
    if not local_map:
        return True
    runs = list(ba_runs_per_stream)
    return len(runs) == n_streams and min(runs) >= 1


def region_is_ba_steady(keyframes_in_region, ba_runs_in_region, local_map):
    """One optimisation per keyframe inside the clock (vo_localmap.cpp:292-366); both counts are taken with the queues drained."""
    if not local_map:
        return True
    return keyframes_in_region > 0 and abs(ba_runs_in_region - keyframes_in_region) <= max(1, keyframes_in_region // 50)


def max_frames(steps, warmup, skip, settle=SETTLE, epilogue=EPILOGUE):
    """Upper bound of the frames a run can feed (for trajectory capacity and the IMU tables)."""
    return frame_schedule(steps, warmup, skip, settle, epilogue, EXTRA_SETTLE_MAX)["n_frames"]


def cpu_sample(requested, sched):
    """Frames of the bounded CPU-baseline sample: the first `requested` frames after the skipped ones, clamped to what the run
    feeds; never empty unless requested <= 0.  Returns (first_frame, count)."""
    if requested <= 0:
        return sched["skip"], 0
    avail = sched["n_frames"] - sched["skip"]
    return sched["skip"], max(1, min(int(requested), avail))


def streams_per_gpu(scaling, world, per_gpu=64, total_strong=512):
    """weak: every rank tracks `per_gpu` streams; strong: a fixed total of `total_strong` streams is split over the ranks."""
    if scaling == "weak":
        return per_gpu
    if scaling == "strong":
        if total_strong % world:
            raise ValueError("strong scaling: %d streams do not split over %d ranks" % (total_strong, world))
        return total_strong // world
    raise ValueError("scaling must be weak or strong")


def percentile(values, q):
    """Nearest-rank percentile (q in [0, 100]) of a non-empty sequence."""
    v = sorted(values)
    if not v:
        raise ValueError("percentile of nothing")
    k = max(0, min(len(v) - 1, int(round(q / 100.0 * (len(v) - 1)))))
    return v[k]
