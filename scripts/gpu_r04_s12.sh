#!/bin/bash
# round 4, session 12: right image in place again (A/B FLVIS_RIGHT_COPY), smaller first tier in k_gftt_pick
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s12
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py -q -m gpu -k "gftt or feature_dem or frontend_parity or cache or features_run_out" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
FLVIS_RIGHT_COPY=1 timeout 300 python bench.py $B < /dev/null > "$OUT/b_rcopy.json" 2> "$OUT/b_rcopy.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_rcopy"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"])
        print("   ", {k: round(v, 4) for k, v in st.items()})
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s12_trace -o b -- python "$R/bench.py" --steps 60 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/trace.log" 2>&1
T=$(find /tmp/s12_trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python "$R/scripts/rocprof_summary.py" "$T" 60 "$OUT/kernel_summary.md" < /dev/null > /dev/null && python "$R/scripts/timeline.py" "$T" < /dev/null > "$OUT/frame_timeline.txt"; head -32 "$OUT/frame_timeline.txt"
