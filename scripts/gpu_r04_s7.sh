#!/bin/bash
# round 4, session 7: how much the local-map workers cost the frame chain (they hold whole CUs: 253 VGPRs x 2 waves per SIMD) -- knobs A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s7
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 400 python bench.py --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
FLVIS_BA_EVERY=2 timeout 300 python bench.py $B < /dev/null > "$OUT/b_every2.json" 2> "$OUT/b_every2.err"
FLVIS_BA_STREAMS=1 timeout 300 python bench.py $B < /dev/null > "$OUT/b_streams1.json" 2> "$OUT/b_streams1.err"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baw4.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_baw4.json" 2> "$OUT/b_baw4.err"
FLVIS_BA_LDS_KB=96 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baw4.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_baw4_lds96.json" 2> "$OUT/b_baw4_lds96.err"
FLVIS_BA_LDS_KB=64 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baw4.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_baw4_lds64.json" 2> "$OUT/b_baw4_lds64.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_every2", "b_streams1", "b_baw4", "b_baw4_lds96", "b_baw4_lds64"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "ba", st.get("ba_worker(launch)"), "tail", ((r.get("latency_ms") or {}).get("timed_region_ms") or {}).get("local_map_tail_after_last_frame"), "h2d", (r.get("with_h2d") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
