#!/bin/bash
# round 4, session 25: phase profile of the local-map solver (-DFLVIS_BA_PROF build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s25
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baprof.so timeout 300 python scripts/ba_prof.py 110 < /dev/null > "$OUT/ba_prof.txt" 2>&1
cat "$OUT/ba_prof.txt"
