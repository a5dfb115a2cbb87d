#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s9
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -x -k "run_out or local_map" < /dev/null > "$OUT/t1.log" 2>&1; tail -3 "$OUT/t1.log"
cd /tmp || exit 1
for i in 1 2; do
for KB in 159 128 112 96; do
  FLVIS_BA_LDS_KB=$KB timeout 200 python "$R/bench.py" --gpus 1 --steps 40 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/b_$KB.json" 2> "$OUT/b_$KB.err"
  python - "$OUT/b_$KB.json" "$KB" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ba = [k for k in r["roofline"]["kernels"] if k["kernel"] == "k_ba_worker"][0]
    print("lds=%s KB  %.1f frames/s  %.4f ms/step  ba ms/opt %.3f  lk_t %.3f lk_s %.3f  kf/ba %s/%s" % (sys.argv[2], r["value"], r["ms_per_step"], ba.get("ms_per_optimisation", -1), r["stages_ms_per_step"]["lk_track(temporal)"], r["stages_ms_per_step"]["lk_track(stereo)"], r["config"]["keyframes_in_timed_region"], r["config"]["ba_runs_in_timed_region"]))
except Exception as e:
    print("lds=%s failed %s" % (sys.argv[2], e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done
done
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/s9_tl -o t -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/tl.json" 2> "$OUT/tl.err"
K=$(find /tmp/s9_tl -name "*kernel_trace.csv" | head -1); M=$(find /tmp/s9_tl -name "*memory_copy_trace.csv" | head -1)
python "$R/scripts/h2d_timeline.py" "$K" "$M" 140 > "$OUT/r03_h2d_timeline.txt"; head -3 "$OUT/r03_h2d_timeline.txt"; tail -45 "$OUT/r03_h2d_timeline.txt"
python - "$OUT/tl.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(r["value"], r.get("with_h2d"))
PY
