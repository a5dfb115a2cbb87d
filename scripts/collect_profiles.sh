#!/bin/bash
# Collects the per-round evidence under profiles/ on the GPU box (run from the repo root, e.g. through gpurun):
#   scripts/collect_profiles.sh rNN
# 1. bench line, 2. rocprofv3 kernel trace of bench.py reduced to a summary, 3. two PMC passes (FETCH_SIZE / WRITE_SIZE,
# separate runs as the microarchitecture guide prescribes) reduced to per-launch HBM traffic of k_lk_track,
# 4. kernel statistics of the ORB path.  Every step has its own timeout and reads nothing from stdin.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout 200 python "$R/bench.py" < /dev/null > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench.err"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_trace -o b -- python "$R/bench.py" --cpu-frames 0 < /dev/null > "$OUT/trace.log" 2>&1
T=$(find /tmp/cp_trace -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/cp_trace -name "*kernel_stats.csv" | head -1)
[ -n "$T" ] && python "$R/scripts/rocprof_summary.py" "$T" 60 "$OUT/${TAG}_bench_kernel_summary.md" < /dev/null > /dev/null
[ -n "$S" ] && grep -E "Name|flvis" "$S" > "$OUT/${TAG}_bench_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cp_pmc_$C -o p -- python "$R/bench.py" --steps 20 --warmup 60 --cpu-frames 0 < /dev/null > "$OUT/pmc_$C.log" 2>&1
done
F=$(find /tmp/cp_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/cp_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python "$R/scripts/pmc_to_json.py" "$F" "$W" k_lk_track "$OUT/${TAG}_lk_pmc.json" < /dev/null > /dev/null
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_orb -o o -- python "$R/scripts/orb_bench.py" 64 10 < /dev/null > "$OUT/${TAG}_orb_bench.log" 2>&1
S=$(find /tmp/cp_orb -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && python "$R/scripts/kernel_stats_brief.py" "$S" < /dev/null > "$OUT/${TAG}_orb_kernel_stats.txt"
ls -la "$OUT"
