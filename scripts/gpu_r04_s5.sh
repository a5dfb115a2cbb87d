#!/bin/bash
# round 4, session 5: whole GPU suite (fused borders, batched tile loads, frame_head per stream), bench, RANSAC / pose LM phase profile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s5
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 1500 python -m pytest tests -q -m gpu < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -8 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
timeout 300 python bench.py $B < /dev/null > "$OUT/b_default.json" 2> "$OUT/b_default.err"
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default",):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {})
        print(n, r["value"], r["ms_per_step"], "chain p50", r["latency_ms"]["gpu_frame_chain_p50"])
        print("   ", {k: round(v, 4) for k, v in st.items()})
    except Exception as e:
        print(n, "failed", e)
PY
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_prof.so timeout 300 python scripts/ba_prof.py 110 < /dev/null > "$OUT/prof.txt" 2>&1; tail -6 "$OUT/prof.txt"
