#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s13
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
run() {
  tag=$1; shift
  env "$@" timeout 200 python "$R/bench.py" --gpus 1 --steps 40 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/b_$tag.json" 2> "$OUT/b_$tag.err"
  python - "$OUT/b_$tag.json" "$tag" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r.get("stages_ms_per_step", {})
    print("%-18s %.1f frames/s  %.4f ms/step  chain p50 %.3f  lanes %s  kf/ba %s/%s" % (sys.argv[2], r["value"], r["ms_per_step"], r["latency_ms"]["gpu_frame_chain_p50"], r["config"]["lanes_per_gpu"], r["config"]["keyframes_in_timed_region"], r["config"]["ba_runs_in_timed_region"]))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
}
run lanes1 FLVIS_LANES=1
run lanes2 FLVIS_LANES=2
run lanes2_nostagger FLVIS_LANES=2 FLVIS_LANE_STAGGER=0
run lanes4 FLVIS_LANES=4
run lanes1_b FLVIS_LANES=1
run lanes2_ba1 FLVIS_LANES=2 FLVIS_BA_STREAMS=1
