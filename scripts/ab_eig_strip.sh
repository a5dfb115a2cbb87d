#!/bin/bash
# Same-session A/B of the opt-in strip-mined corner-response kernel (FLVIS_EIG_STRIP=1, k_eig_cand_strip) against the default
# k_eig_cand, on the GPU box (run from the repo root through gpurun; about 2.5 minutes):
#   scripts/ab_eig_strip.sh rNN
# 1. parity first: the goodFeaturesToTrack / FeatureDEM tests with the switch on; 2. the bench line with the driver's arguments,
# default / strip / default / strip (alternating, so that a drift of the box shows up as a difference between equal runs);
# 3. per-kernel times of both from one rocprofv3 kernel trace each.  Results under gpurun_out/ab_eig_strip_rNN/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/ab_eig_strip_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
FLVIS_EIG_WALK=0 FLVIS_EIG_STRIP=1 FLVIS_EIG_STRIP_CHILD=1 timeout 120 python -m pytest tests/test_gpu_image.py -q -k "gftt or feature_dem" < /dev/null > "$OUT/parity.log" 2>&1
tail -2 "$OUT/parity.log"
cd /tmp || exit 1
for i in 1 2; do
  for V in 0 1; do
    FLVIS_EIG_WALK=0 FLVIS_EIG_STRIP=$V timeout 200 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null \
      > "$OUT/bench_strip${V}_run$i.json" 2> "$OUT/bench_strip${V}_run$i.err"
    python - "$OUT/bench_strip${V}_run$i.json" "$V" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("strip=%s  %.1f frames/s  %.4f ms/step  eig_cand stage %.4f ms" % (sys.argv[2], r["value"], r["ms_per_step"], r["stages_ms_per_step"]["gftt:eig_cand"]))
except Exception as e:
    print("strip=%s  no bench line (%s)" % (sys.argv[2], e))
PY
  done
done
for V in 0 1; do
  FLVIS_EIG_WALK=0 FLVIS_EIG_STRIP=$V timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_trace$V -o b -- python "$R/bench.py" --steps 40 --warmup 10 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/trace$V.log" 2>&1
  S=$(find /tmp/ab_trace$V -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && grep -E "Name|k_eig_cand|k_lk_track|k_gftt_pick" "$S" > "$OUT/kernel_stats_strip$V.csv" && cat "$OUT/kernel_stats_strip$V.csv"
done
