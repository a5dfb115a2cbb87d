#!/bin/bash
# round 4, session 32: the left pyramid's second launch on the main stream behind k_frame_head_prepare (FLVIS_PYR_SPLIT) against both on the detection stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s32
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "frontend or cache or kitti or euroc or cpp_caller or config or starvation or run_out" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -3 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run split1 FLVIS_NOP=1
run both1 FLVIS_PYR_SPLIT=0
run split2 FLVIS_NOP=1
run both2 FLVIS_PYR_SPLIT=0
run split3 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        st = r.get("stages_ms_per_step", {}) or {}
        print(os.path.basename(f), r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"), "pyr", st.get("pyr_down(left)"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
