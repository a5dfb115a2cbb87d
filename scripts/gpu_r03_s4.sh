#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s4
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_image.py -q -x < /dev/null > "$OUT/t_image.log" 2>&1; tail -5 "$OUT/t_image.log"
timeout 600 python -m pytest tests/test_gpu_dataset_runner.py tests/test_gpu_pipeline.py -q -x < /dev/null > "$OUT/t_pipe.log" 2>&1; tail -5 "$OUT/t_pipe.log"
cd /tmp || exit 1
for i in 1 2; do
  for V in 0 60 120 240; do
    if [ "$V" = 0 ]; then unset FLVIS_EIG_WALK; else export FLVIS_EIG_WALK=$V; fi
    timeout 200 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/bench_walk${V}_run$i.json" 2> "$OUT/bench_walk${V}_run$i.err"
    python - "$OUT/bench_walk${V}_run$i.json" "$V" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st = r["stages_ms_per_step"]
    print("walk=%s  %.1f frames/s  %.4f ms/step  eig_cand %.4f  lk(temporal) %.4f" % (sys.argv[2], r["value"], r["ms_per_step"], st["gftt:eig_cand"], st.get("lk_track(temporal)", -1)))
except Exception as e:
    print("walk=%s  no bench line (%s)" % (sys.argv[2], e))
PY
  done
done
unset FLVIS_EIG_WALK
