#!/bin/bash
# SQ counters of the LK launches (one rocprofv3 --pmc pass, SQ block only): where a wave's cycles go
#   scripts/lk_pmc.sh <outdir> [env assignments for the bench...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for ENVV in "$@"; do export "$ENVV"; done
B="--steps 20 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue --no-h2d"
rm -rf /tmp/lkpmc1 /tmp/lkpmc2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/lkpmc1 -o p -- python "$R/bench.py" $B < /dev/null > "$OUT/pmc1.log" 2>&1
F=$(find /tmp/lkpmc1 -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python "$R/scripts/pmc_summary.py" "$F" > "$OUT/sq_counters_1.txt" && head -12 "$OUT/sq_counters_1.txt"
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d /tmp/lkpmc2 -o p -- python "$R/bench.py" $B < /dev/null > "$OUT/pmc2.log" 2>&1
F=$(find /tmp/lkpmc2 -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python "$R/scripts/pmc_summary.py" "$F" > "$OUT/sq_counters_2.txt" && head -12 "$OUT/sq_counters_2.txt"
