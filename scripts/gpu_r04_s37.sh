#!/bin/bash
# round 4, session 37: one bench line at the head (after the last local-map change), tracking part only
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s37
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 100 python bench.py --no-h2d --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/r04_bench_line_head_tracking_only.json" 2> "$OUT/b.err"
python - "$OUT" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1] + "/r04_bench_line_head_tracking_only.json").read().strip().splitlines()[-1])
l = r["latency_ms"]
print(r["value"], r["ms_per_step"], l["gpu_frame_chain_p50"], l["gpu_frame_chain_p99"], l["timed_region_ms"])
PY
