#!/bin/bash
# Collects the per-round evidence under profiles/ on the GPU box (run from the repo root, e.g. through gpurun):
#   scripts/collect_profiles.sh rNN
# 1. the bench line with the driver's arguments and with the defaults, 2. rocprofv3 kernel trace of bench.py reduced to a summary
# + the kernel timeline of one frame, 3. `bench.py --pmc` (two separate PMC passes: FETCH_SIZE / WRITE_SIZE of k_lk_track),
# 4. MFMA counters of k_ba_worker with the register-tile and the MFMA Schur variant.  Every step has its own timeout.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout 300 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 < /dev/null > "$OUT/${TAG}_bench_line_driver_args.json" 2> "$OUT/bench_driver.err"
timeout 300 python "$R/bench.py" < /dev/null > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_trace -o b -- python "$R/bench.py" --steps 60 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/trace.log" 2>&1
T=$(find /tmp/cp_trace -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/cp_trace -name "*kernel_stats.csv" | head -1)
[ -n "$T" ] && python "$R/scripts/rocprof_summary.py" "$T" 60 "$OUT/${TAG}_bench_kernel_summary.md" < /dev/null > /dev/null
[ -n "$T" ] && python "$R/scripts/timeline.py" "$T" < /dev/null > "$OUT/${TAG}_frame_timeline.txt"
[ -n "$S" ] && grep -E "Name|flvis" "$S" > "$OUT/${TAG}_bench_kernel_stats.csv"
timeout 600 python "$R/bench.py" --pmc --steps 20 --warmup 10 < /dev/null > "$OUT/pmc.log" 2>&1
cp "$R/gpurun_out/${TAG}_lk_pmc.json" "$OUT/" 2>/dev/null
for M in 0 1; do
  FLVIS_BA_MFMA=$M timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/cp_mfma$M -o m -- python "$R/bench.py" --steps 20 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/mfma_pmc_$M.log" 2>&1
  F=$(find /tmp/cp_mfma$M -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$R/scripts/pmc_summary.py" "$F" < /dev/null > "$OUT/${TAG}_ba_mfma_counters_variant$M.txt"
done
ls -la "$OUT"
