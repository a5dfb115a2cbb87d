#!/bin/bash
# round 4, session 28: keyframes a local-map workgroup takes per launch (FLVIS_BA_DRAIN; 0 = until the queue is empty) with the deeper queue
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s28
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "local_map or config or feedback or cpp_caller" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -4 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
for i in 1 2 3 4; do
  run d2_$i FLVIS_BA_DRAIN=2
  run d0_$i FLVIS_BA_DRAIN=0
  run d1_$i FLVIS_BA_DRAIN=1
  run d3_$i FLVIS_BA_DRAIN=3
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        st = r.get("stages_ms_per_step", {}) or {}
        print(os.path.basename(f), r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "ba launch", st.get("ba_worker(launch)"),
              "tail", (l.get("timed_region_ms") or {}).get("local_map_tail_after_last_frame"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"))
    except Exception as e:
        print(os.path.basename(f), "failed", e, open(f.replace(".json", ".err")).read()[-300:])
PY
