#!/bin/bash
# round 4, session 23: where the occasional 5-8 ms frame chains come from (keyframe back-pressure?): local-map streams 2 / 3 / 4, a launch
# every second frame, three runs each with the per-frame chain times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s23
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
for i in 1 2 3; do
  run s2_$i FLVIS_NOP=1
  run s3_$i FLVIS_BA_STREAMS=3
  run s4_$i FLVIS_BA_STREAMS=4
  run e2_$i FLVIS_BA_EVERY=2
done
run nolm_1 FLVIS_NOP=1
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
        print(f, "failed", e)
PY
