#!/bin/bash
# round 4, session 3: LK with all staging loads in flight, slot selection hoisted, header check in k_track_prepare, next-level prefetch (A/B)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s3
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py -q -m gpu -k "lk or frontend_parity_two or frontend_parity_euroc_mode or kitti or cache" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_nopf.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_nopf.json" 2> "$OUT/b_nopf.err"
FLVIS_LK_TCACHE=0 timeout 300 python bench.py $B < /dev/null > "$OUT/b_notc.json" 2> "$OUT/b_notc.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_nopf", "b_notc"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"], "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "pyrL", st.get("pyr_down(left)"), "prep", st.get("track_prepare"))
    except Exception as e:
        print(n, "failed", e)
PY
bash scripts/lk_pmc.sh "$OUT/pmc_default"
