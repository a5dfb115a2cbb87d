#!/bin/bash
# round 4, session 6: auxiliary stream (vi correction + triangulation behind the filter), lanes A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s6
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "frontend_parity or config3 or config2_single" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
FLVIS_LANES=2 timeout 300 python bench.py $B < /dev/null > "$OUT/b_lanes2.json" 2> "$OUT/b_lanes2.err"
FLVIS_LANES=2 FLVIS_BA_LDS_KB=96 timeout 300 python bench.py $B < /dev/null > "$OUT/b_lanes2_lds96.json" 2> "$OUT/b_lanes2_lds96.err"
FLVIS_LANES=2 timeout 300 python bench.py $B --no-local-map < /dev/null > "$OUT/b_lanes2_nolm.json" 2> "$OUT/b_lanes2_nolm.err"
timeout 300 python bench.py $B --no-local-map < /dev/null > "$OUT/b_nolm.json" 2> "$OUT/b_nolm.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_lanes2", "b_lanes2_lds96", "b_lanes2_nolm", "b_nolm"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "dem", st.get("feature_dem+add_new"), "innov", st.get("depth_innovate"), "tail", ((r.get("latency_ms") or {}).get("timed_region_ms") or {}).get("local_map_tail_after_last_frame"))
    except Exception as e:
        print(n, "failed", e)
PY
