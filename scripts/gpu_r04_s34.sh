#!/bin/bash
# round 4, session 34: the per-frame input block read in place from page-locked host memory (FLVIS_INPUT_ZEROCOPY) against the H2D copy
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s34
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_dataset_runner.py -q -m gpu -k "frontend or imu or depth or kitti or euroc or cpp_caller or config or host or keyframe_msg or dataset or staging" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -3 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run zc1 FLVIS_NOP=1
run copy1 FLVIS_INPUT_ZEROCOPY=0
run zc2 FLVIS_NOP=1
run copy2 FLVIS_INPUT_ZEROCOPY=0
run zc3 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        st = r.get("stages_ms_per_step", {}) or {}
        t = l.get("timed_region_ms") or {}
        print(os.path.basename(f), r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "between frames", round(t.get("tracking_stream_span", 0) - t.get("sum_of_frame_chains", 0), 2), "head", st.get("imu_feed+frame_begin"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
