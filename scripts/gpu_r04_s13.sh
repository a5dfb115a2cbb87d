#!/bin/bash
# round 4, session 13: right pyramid beside the F-RANSAC (FLVIS_DET_START=4), rows per chunk of the corner-response walk
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s13
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run det3 FLVIS_DET_START=3
run det4 FLVIS_DET_START=4
run det4_w60 FLVIS_DET_START=4 FLVIS_EIG_WALK=60
run det4_w40 FLVIS_DET_START=4 FLVIS_EIG_WALK=40
run det4_w240 FLVIS_DET_START=4 FLVIS_EIG_WALK=240
python - "$OUT" <<'PY'
import json, sys
for n in ("b_det3", "b_det4", "b_det4_w60", "b_det4_w40", "b_det4_w240"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"], "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"), "f", st.get("ransac_f"), "eig", st.get("gftt:eig_cand"), "pick", st.get("gftt:pick"), "dem", st.get("feature_dem+add_new"))
    except Exception as e:
        print(n, "failed", e)
PY
