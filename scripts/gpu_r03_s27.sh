#!/bin/bash
# LK staging fast paths: LK parity (bit-exact positions / flags), the closed loop, the bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s27
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_image.py tests/test_golden.py -x -q -m gpu -k "lk or golden or fixture" > "$OUT/t1.log" 2>&1; tail -2 "$OUT/t1.log"
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_loop.py -x -q -m gpu -k "frontend_parity or lc_keyframe or verification" > "$OUT/t2.log" 2>&1; tail -2 "$OUT/t2.log"
cd /tmp
for i in 1 2; do
timeout 200 python "$R/bench.py" --gpus 1 --steps 60 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
python - "$OUT/b_$i.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r.get("stages_ms_per_step", {})
print("%.1f frames/s  %.4f ms/step  chain p50 %.3f " % (r["value"], r["ms_per_step"], r["latency_ms"]["gpu_frame_chain_p50"]), {k: round(v, 3) for k, v in st.items() if v > 0.02})
PY
done
