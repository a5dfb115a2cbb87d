#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s18
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baprof.so timeout 300 python "$R/scripts/ba_prof.py" 140 > "$OUT/prof.log" 2>&1; cat "$OUT/prof.log" | tail -22
