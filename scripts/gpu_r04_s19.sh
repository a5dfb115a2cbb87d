#!/bin/bash
# round 4, session 19: the pyramid kernels alone (scripts/pyr_bench.py under a kernel trace), band sizes, tile kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s19
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
one() { n=$1; shift; rm -rf /tmp/s19_$n; env "$@" PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s19_$n -o b -- python "$R/scripts/pyr_bench.py" 64 40 ${ING:-1} < /dev/null > "$OUT/$n.log" 2>&1
  S=$(find /tmp/s19_$n -name "*kernel_stats.csv" | head -1); echo "== $n"; [ -n "$S" ] && grep -E "pyr" "$S" | cut -d, -f1-6 | tee "$OUT/$n.csv"; }
one walk FLVIS_NOP=1
one tiles FLVIS_PYR_TILES=1
one band2 FLVIS_PYR_BAND=2 FLVIS_PYR_BAND2=1
one band8 FLVIS_PYR_BAND=8 FLVIS_PYR_BAND2=4
one plan21 FLVIS_PYR_PLAN=21
ING=0 one walk_noingest FLVIS_NOP=1
ING=0 one tiles_noingest FLVIS_PYR_TILES=1
