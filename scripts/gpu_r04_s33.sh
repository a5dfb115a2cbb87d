#!/bin/bash
# round 4, session 33: the pyramid plan chosen by the image width (752-pixel rows: two levels first) -- parity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s33
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 600 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py -q -m gpu -k "pyr or euroc or frontend_parity" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -3 "$OUT/gpu_tests.log"
