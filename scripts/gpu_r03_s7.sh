#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s7
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_loop.py tests/test_gpu_loop_closer.py -q -x < /dev/null > "$OUT/t1.log" 2>&1; tail -15 "$OUT/t1.log"
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_golden.py tests/test_cpp_caller.py tests/test_gpu_dataset_runner.py -q -x -m gpu < /dev/null > "$OUT/t2.log" 2>&1; tail -8 "$OUT/t2.log"
cd /tmp || exit 1
timeout 300 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], {k: v for k, v in r["stages_ms_per_step"].items() if "ransac" in k or "lk" in k})
PY
