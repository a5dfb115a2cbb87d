#!/bin/bash
# round 4, session 2: the whole GPU suite (template cache fixed, FeatureDEM tie order), A/B bench lines, calibration table, S = 1 latency
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s2
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 1500 python -m pytest tests -q -m gpu < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -25 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
FLVIS_LK_TCACHE=0 timeout 300 python bench.py $B < /dev/null > "$OUT/b_notc.json" 2> "$OUT/b_notc.err"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_lkw5.so timeout 300 python bench.py $B < /dev/null > "$OUT/b_w5.json" 2> "$OUT/b_w5.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_notc", "b_w5"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"], "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "pyrL", st.get("pyr_down(left)"), r["roofline"].get("lk_iterations", {}).get("template_cache"))
        print("   ", {k: round(v, 4) for k, v in st.items()})
    except Exception as e:
        print(n, "failed", e)
PY
timeout 600 python scripts/pmc_calibrate.py r04 "$OUT" < /dev/null > "$OUT/calibrate.log" 2>&1; tail -14 "$OUT/calibrate.log"
timeout 600 python scripts/s1_latency.py "$OUT/r04_s1_latency.json" < /dev/null > "$OUT/s1.log" 2>&1; tail -2 "$OUT/s1.log"
