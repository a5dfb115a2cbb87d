#!/bin/bash
# round 4, session 20: the per-frame input block uploaded early on the detection stream (device ring) against the upload on the main stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s20
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_dataset_runner.py -q -m gpu -k "frontend or imu or kitti or euroc or cpp_caller or config or host or dataset or keyframe_msg" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -6 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run early FLVIS_NOP=1
run main FLVIS_INPUT_UPLOAD=0
run early2 FLVIS_NOP=1
run main2 FLVIS_INPUT_UPLOAD=0
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_driver.json" 2> "$OUT/b_driver.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_early", "b_main", "b_early2", "b_main2", "b_driver"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "h2d", (r.get("with_h2d") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
