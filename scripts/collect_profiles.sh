#!/bin/bash
# Collects the per-round evidence on the GPU box (run from the repo root through gpurun; copy what it leaves under
# gpurun_out/profiles_<tag>/ into profiles/ afterwards):
#   scripts/collect_profiles.sh rNN
# 1. `bench.py --pmc`: four separate rocprofv3 passes of the bench itself (kernel trace; SQ_INSTS_VALU; FETCH_SIZE; WRITE_SIZE) ->
#    <tag>_kernel_pmc.json + <tag>_lk_pmc.json, which the bench lines below read back (source hash checked);
# 2. the bench line with the driver's arguments and with the defaults;
# 3. rocprofv3 --kernel-trace --stats of bench.py reduced to a summary + the kernel timeline of one frame.
# Every step has its own timeout.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout 900 python "$R/bench.py" --pmc --steps 20 --warmup 10 < /dev/null > "$OUT/pmc.log" 2>&1
cp "$R/gpurun_out/${TAG}_lk_pmc.json" "$R/gpurun_out/${TAG}_kernel_pmc.json" "$OUT/" 2>/dev/null
timeout 300 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 < /dev/null > "$OUT/${TAG}_bench_line_driver_args.json" 2> "$OUT/bench_driver.err"
timeout 400 python "$R/bench.py" < /dev/null > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_trace -o b -- python "$R/bench.py" --steps 60 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/trace.log" 2>&1
T=$(find /tmp/cp_trace -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/cp_trace -name "*kernel_stats.csv" | head -1)
[ -n "$T" ] && python "$R/scripts/rocprof_summary.py" "$T" 60 "$OUT/${TAG}_bench_kernel_summary.md" < /dev/null > /dev/null
[ -n "$T" ] && python "$R/scripts/timeline.py" "$T" < /dev/null > "$OUT/${TAG}_frame_timeline.txt"
[ -n "$S" ] && grep -E "Name|flvis" "$S" > "$OUT/${TAG}_bench_kernel_stats.csv"
python - "$OUT" "$TAG" <<'PY'
import json, sys
for n in ("_bench_line_driver_args", "_bench_line"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + sys.argv[2] + n + ".json").read().strip().splitlines()[-1])
        c = r["config"]
        print(n, r["value"], r["ms_per_step"], "kf/ba in region", c["keyframes_in_timed_region"], c["ba_runs_in_timed_region"], "h2d", r.get("with_h2d", {}).get("value"),
              "cpu", r.get("cpu_baseline", {}).get("value"), "roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "traffic")})
    except Exception as e:
        print(n, "failed", e)
PY
ls -la "$OUT"
