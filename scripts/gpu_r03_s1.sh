#!/bin/bash
# Round-3 GPU session 1: same-session A/B of the strip-mined corner response + its SQ_INSTS_VALU count.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
bash scripts/ab_eig_strip.sh r03
OUT=$R/gpurun_out/ab_eig_strip_r03
export TMPDIR=/tmp
cd /tmp || exit 1
for V in 0 1; do
  FLVIS_EIG_STRIP=$V timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/s1_pmc$V -o m -- python "$R/bench.py" --steps 20 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/pmc_$V.log" 2>&1
  F=$(find /tmp/s1_pmc$V -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$R/scripts/pmc_summary.py" "$F" < /dev/null > "$OUT/valu_counters_strip$V.txt"
  head -8 "$OUT/valu_counters_strip$V.txt"
done
