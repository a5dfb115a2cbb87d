"""bench.py's frame schedule (pure functions) and its rank-spawning + result-exchange path (gloo, world size 2, GPU work
stubbed by --stub).  Round 1's driver run `--steps 20 --warmup 5` timed nothing but skipped start-up frames and crashed in
the CPU leg; these tests pin the schedule for exactly those arguments."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flvis_amd import bench_plan as plan  # noqa: E402


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1, 0), (60, 70), (60, 10)])
@pytest.mark.parametrize("skip", [50, 0])
def test_timed_region_is_always_after_skip_and_settle(steps, warmup, skip):
    s = plan.frame_schedule(steps, warmup, skip)
    t0, t1 = s["timed"]
    assert t1 - t0 == steps                                  # EXACTLY K timed steps
    assert s["warmup"][1] - s["warmup"][0] == warmup
    assert t0 >= skip + plan.SETTLE                          # never a skipped / init frame inside the clock
    assert s["preroll"] == (0, skip + plan.SETTLE)
    assert s["warmup"][0] == s["preroll"][1] and s["timed"][0] == s["warmup"][1] and s["epilogue"][0] == t1
    assert s["n_frames"] == t1 + plan.EPILOGUE
    # ranges are contiguous and disjoint
    covered = []
    for k in ("preroll", "warmup", "timed", "epilogue"):
        covered += list(range(*s[k]))
    assert covered == list(range(s["n_frames"]))
    assert plan.max_frames(steps, warmup, skip) == s["n_frames"] + plan.EXTRA_SETTLE_MAX


def test_extra_settle_shifts_everything():
    a = plan.frame_schedule(20, 5, 50)
    b = plan.frame_schedule(20, 5, 50, extra_settle=3)
    for k in ("warmup", "timed", "epilogue"):
        assert b[k] == (a[k][0] + 3, a[k][1] + 3)
    assert b["preroll"] == (0, a["preroll"][1] + 3)


def test_preroll_ends_only_in_the_ba_steady_state():
    """The local map optimises only once a stream's window is full (vo_localmap.cpp:211-214): the pre-roll goes on until every stream
    tracks AND has run an optimisation, and a value is printed only if the timed region saw one optimisation per keyframe."""
    assert not plan.steady_state(63, 64, [1] * 64, True)                 # a stream not tracking yet
    assert not plan.steady_state(64, 64, [1] * 63 + [0], True)           # a window still filling
    assert not plan.steady_state(64, 64, [], True)
    assert plan.steady_state(64, 64, [1] * 63 + [3], True)
    assert plan.steady_state(64, 64, [], False)                          # front-end only runs: tracking is enough
    assert plan.region_is_ba_steady(412, 412, True) and plan.region_is_ba_steady(412, 410, True)
    assert not plan.region_is_ba_steady(883, 435, True)                  # round 2's driver line: half the keyframes never optimised
    assert not plan.region_is_ba_steady(0, 0, True)
    assert plan.region_is_ba_steady(0, 0, False)
    assert plan.EXTRA_SETTLE_MAX >= 60                                   # room for the ~40 tracked frames a window needs to fill


def test_cpu_sample_is_never_empty_for_the_driver_arguments():
    s = plan.frame_schedule(20, 5, 50)
    first, n = plan.cpu_sample(80, s)
    assert first == 50 and 1 <= n <= 80 and first + n <= s["n_frames"]
    assert plan.cpu_sample(80, plan.frame_schedule(1, 0, 50, epilogue=0)) == (50, plan.SETTLE + 1)
    assert plan.cpu_sample(0, s) == (50, 0)
    assert plan.cpu_sample(5, s) == (50, 5)


def test_bad_arguments_fail_loudly():
    with pytest.raises(ValueError):
        plan.frame_schedule(0, 5, 50)
    with pytest.raises(ValueError):
        plan.frame_schedule(5, -1, 50)
    with pytest.raises(ValueError):
        plan.streams_per_gpu("strong", 3, 64, 512)
    with pytest.raises(ValueError):
        plan.streams_per_gpu("sideways", 1)


def test_weak_and_strong_scaling_stream_counts():
    assert [plan.streams_per_gpu("weak", n) for n in (1, 2, 4, 8)] == [64, 64, 64, 64]
    assert [plan.streams_per_gpu("strong", n) for n in (1, 2, 4, 8)] == [512, 256, 128, 64]


def test_percentile():
    v = list(range(1, 101))
    assert plan.percentile(v, 50) in (50, 51) and plan.percentile(v, 99) == 99 and plan.percentile([7], 99) == 7


def _run_bench(extra, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_bench_spawns_two_ranks_and_exchanges_results_on_gloo():
    """`bench.py --gpus 2 --stub`: self-spawn through torch.distributed.run, shard, barrier, max-over-ranks, all-gather /
    all-reduce -- everything of the N>1 path except the GPU work."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--streams", "4"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stub"] is True and d["value"] is None and d["steps"] == 3
    assert d["config"]["streams_total"] == 8 and d["config"]["frames_exchanged"] == 2 * 4 * 3
    assert d["ms_per_step"] > 0


def test_bench_stub_at_eight_ranks():
    """BASELINE.json configs[4] has 8 ranks on one node: the spawn / shard / barrier / exchange path at world size 8 (gloo, no GPU work):
    512 streams in all, global stream ids in order after the all-gather (asserted inside run_stub), one JSON line."""
    r = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "0", "--stub"], timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["streams_per_gpu"] == 64 and d["config"]["streams_total"] == 512
    assert d["config"]["frames_exchanged"] == 8 * 64 * 2 and d["config"]["streams_tracking_at_end"] == 512


def test_bench_stub_strong_scaling_splits_a_fixed_total():
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub", "--scaling", "strong", "--total-streams", "16"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["streams_per_gpu"] == 8 and d["config"]["streams_total"] == 16


def test_bench_refuses_more_ranks_than_gpus():
    """Without --stub, `--gpus 2` on a box with fewer than 2 GPUs must fail loudly, not print an n_gpus: 1 line."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"])
    assert r.returncode != 0
    assert b"GPU(s) visible" in r.stderr and not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stderr
