#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s17
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 600 python -m pytest tests/test_gpu_epnp.py -x -q -m gpu > "$OUT/t0.log" 2>&1; grep -n "^E  \|passed\|failed" "$OUT/t0.log" | head -10
timeout 600 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "pnp_ransac or verification_chain" > "$OUT/t1.log" 2>&1; tail -3 "$OUT/t1.log"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "frontend_parity" > "$OUT/t2.log" 2>&1; tail -3 "$OUT/t2.log"
cd /tmp
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_prof.so timeout 300 python "$R/scripts/ba_prof.py" 110 > "$OUT/prof.log" 2>&1; tail -4 "$OUT/prof.log" | grep -v seven
timeout 300 python "$R/bench.py" --gpus 1 --steps 60 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r.get("stages_ms_per_step", {})
print(r["value"], r["ms_per_step"], {k: st[k] for k in ("ransac_f", "ransac_pnp", "track_post+pose_lm") if k in st})
PY
