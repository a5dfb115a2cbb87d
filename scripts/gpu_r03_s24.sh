#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s24
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
echo "== sums of the final EPnP on a few lanes (previous commit)"; FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_oldepnp.so timeout 300 python scripts/epnp_bench.py 2>&1 | tail -2
echo "== chunked sums on the whole workgroup"; timeout 300 python scripts/epnp_bench.py 2>&1 | tail -2
