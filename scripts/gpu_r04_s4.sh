#!/bin/bash
# round 4, session 4: temporal LK with the template computation out of line (register budget A/B), k_frame_head one wave per stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s4
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py -q -m gpu -k "lk or frontend_parity or cache or imu" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
for V in lkt5 lkt6 lkt8; do
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_$V.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_$V.json" 2> "$OUT/b_$V.err"
done
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_lkt5", "b_lkt6", "b_lkt8"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"], "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "head", st.get("imu_feed+frame_begin"), "pyrL", st.get("pyr_down(left)"))
    except Exception as e:
        print(n, "failed", e)
PY
