#!/bin/bash
# round 4, session 14: the host-image leg -- every kernel and copy of its last frames (where does a frame wait?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s14
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/s14_trace -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/bench_h2d.json" 2> "$OUT/bench_h2d.err"
K=$(find /tmp/s14_trace -name "*kernel_trace.csv" | head -1)
M=$(find /tmp/s14_trace -name "*memory_copy_trace.csv" | head -1)
python "$R/scripts/h2d_full_timeline.py" "$K" "$M" 3 > "$OUT/h2d_full_timeline.txt" 2>&1
python - "$OUT/bench_h2d.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r.get("with_h2d"))
PY
head -75 "$OUT/h2d_full_timeline.txt"
