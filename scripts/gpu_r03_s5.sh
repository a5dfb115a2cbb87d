#!/bin/bash
# bench with the BA-steady-state pre-roll (driver args + defaults), loop tests, VALU counters of the default build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s5
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_image.py -q -x -k "reject_bad or other_response or strip_mined" < /dev/null > "$OUT/t.log" 2>&1; tail -3 "$OUT/t.log"
cd /tmp || exit 1
timeout 300 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 < /dev/null > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; tail -c 600 "$OUT/bench_driver.err"
timeout 300 python "$R/bench.py" < /dev/null > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("bench_driver", "bench_default"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        c = r["config"]
        print(n, r["value"], r["ms_per_step"], "preroll", c["preroll_frames"], "kf/ba in region", c["keyframes_in_timed_region"], c["ba_runs_in_timed_region"], "h2d", r.get("with_h2d", {}).get("value"), "cpu", r.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/s5_pmc -o m -- python "$R/bench.py" --steps 20 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/pmc.log" 2>&1
F=$(find /tmp/s5_pmc -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python "$R/scripts/pmc_summary.py" "$F" < /dev/null > "$OUT/valu_counters.txt"; head -12 "$OUT/valu_counters.txt"
