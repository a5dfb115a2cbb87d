#!/bin/bash
# round 4, session 11: where the detection stream starts (FLVIS_DET_START), now that the temporal LK and the pose LM are shorter
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s11
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
for M in 3 2 1 0; do
FLVIS_DET_START=$M timeout 300 python bench.py $B < /dev/null > "$OUT/b_det$M.json" 2> "$OUT/b_det$M.err"
done
python - "$OUT" <<'PY'
import json, sys
for n in ("b_det3", "b_det2", "b_det1", "b_det0"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"], "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"), "f", st.get("ransac_f"), "pnp", st.get("ransac_pnp"), "lm", st.get("track_post+pose_lm"), "dem", st.get("feature_dem+add_new"))
    except Exception as e:
        print(n, "failed", e)
PY
