#!/bin/bash
# the whole GPU suite once more (log for profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r03
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 2400 python -m pytest tests -q -m gpu < /dev/null > "$OUT/r03_gpu_tests.log" 2>&1; tail -5 "$OUT/r03_gpu_tests.log"
