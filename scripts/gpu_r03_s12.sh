#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s12
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py -q -x -k "lk or two_streams or euroc_mode or kitti" < /dev/null > "$OUT/t1.log" 2>&1; tail -3 "$OUT/t1.log"
cd /tmp || exit 1
for i in 1 2; do
  timeout 200 python "$R/bench.py" --gpus 1 --steps 40 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
  python - "$OUT/b_$i.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r["stages_ms_per_step"]
print("%.1f frames/s  %.4f ms/step  lk_t %.3f lk_s %.3f  chain p50 %.3f" % (r["value"], r["ms_per_step"], st["lk_track(temporal)"], st["lk_track(stereo)"], r["latency_ms"]["gpu_frame_chain_p50"]))
PY
done
