#!/bin/bash
# chunked sums in EPnP: parity (EPnP alone, RANSAC sets, loop closing, the four rigs), timing of the standalone kernel and of a KITTI-like run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s23
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_epnp.py tests/test_gpu_loop.py tests/test_gpu_loop_closer.py -x -q -m gpu > "$OUT/t1.log" 2>&1; tail -3 "$OUT/t1.log"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "frontend_parity" --durations=8 > "$OUT/t2.log" 2>&1; tail -14 "$OUT/t2.log"
timeout 300 python scripts/epnp_bench.py > "$OUT/epnp_bench.log" 2>&1; tail -3 "$OUT/epnp_bench.log"
