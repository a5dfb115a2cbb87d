#!/bin/bash
# where the detection stream's work runs: beside the temporal LK (0), after it (1), after the F-RANSAC (2), right pyramid too (3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s19
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {
  tag=$1; shift
  env "$@" timeout 200 python "$R/bench.py" --gpus 1 --steps 60 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-h2d < /dev/null > "$OUT/b_$tag.json" 2> "$OUT/b_$tag.err"
  python - "$OUT/b_$tag.json" "$tag" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); st = r.get("stages_ms_per_step", {})
    print("%-14s %.1f frames/s  %.4f ms/step  chain p50 %.3f " % (sys.argv[2], r["value"], r["ms_per_step"], r["latency_ms"]["gpu_frame_chain_p50"]),
          {k: st[k] for k in ("lk_track(temporal)", "gftt:eig_cand", "gftt:pick", "ransac_f", "ransac_pnp", "track_post+pose_lm", "reproj_filter", "feature_dem+add_new", "lk_track(stereo)", "depth_innovate")})
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
}
run after_f FLVIS_DET_START=2
run pyr_late FLVIS_DET_START=3
run after_f_b FLVIS_DET_START=2
run pyr_late_b FLVIS_DET_START=3
