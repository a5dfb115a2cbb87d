#!/bin/bash
# round 4, session 15: host-image uploads gated under the LK launches: tests that use flvis_image_feed_host, the with_h2d leg, its timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s15
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -q -m gpu -k "host or cpp_caller or dataset or depth_camera or kitti or keyframe_msg" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/s15_trace -o b -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/bench_h2d.json" 2> "$OUT/bench_h2d.err"
K=$(find /tmp/s15_trace -name "*kernel_trace.csv" | head -1)
M=$(find /tmp/s15_trace -name "*memory_copy_trace.csv" | head -1)
python "$R/scripts/h2d_full_timeline.py" "$K" "$M" 3 > "$OUT/h2d_full_timeline.txt" 2>&1
timeout 400 python "$R/bench.py" --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("bench_h2d", "bench_default"):
    r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1]); print(n, r["value"], r["ms_per_step"], r.get("with_h2d"))
PY
head -36 "$OUT/h2d_full_timeline.txt"
