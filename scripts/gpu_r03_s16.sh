#!/bin/bash
# EPnP hypotheses on 4 waves (one per SIMD) instead of 8: phase times and stage time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s16
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_prof4.so timeout 300 python "$R/scripts/ba_prof.py" 110 > "$OUT/prof.log" 2>&1; tail -4 "$OUT/prof.log" | grep -v seven
for v in w4 ""; do
FLVIS_LIB_PATH=${v:+$R/build_variants/libflvis_hip_$v.so} timeout 300 python "$R/bench.py" --gpus 1 --steps 60 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
python - "$OUT/bench_$v.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r.get("stages_ms_per_step", {})
print(sys.argv[1][-12:], r["value"], r["ms_per_step"], {k: st[k] for k in ("ransac_f", "ransac_pnp", "track_post+pose_lm") if k in st})
PY
done
