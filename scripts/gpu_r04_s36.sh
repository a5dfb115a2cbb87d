#!/bin/bash
# round 4, session 36 (last of the round): local-map and 64-stream configuration tests at the head (two-pass Schur accumulate)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s36
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 150 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "local_map or config" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -3 "$OUT/gpu_tests.log"
