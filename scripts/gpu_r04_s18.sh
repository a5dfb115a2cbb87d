#!/bin/bash
# round 4, session 18: the walking pyramid kernels (pyr_walk.hip) -- parity, then A/B against the tile kernels (FLVIS_PYR_TILES=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s18
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_image.py -q -m gpu -k "pyr" < /dev/null > "$OUT/gpu_tests_pyr.log" 2>&1; tail -15 "$OUT/gpu_tests_pyr.log"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "frontend or cache or kitti or euroc or cpp_caller or config" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -6 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run walk FLVIS_NOP=1
run tiles FLVIS_PYR_TILES=1
run walk21 FLVIS_PYR_PLAN=21
run walk_b4 FLVIS_PYR_BAND=4
run walk_b16 FLVIS_PYR_BAND=16
run walk2 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys
for n in ("b_walk", "b_tiles", "b_walk21", "b_walk_b4", "b_walk_b16", "b_walk2"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              {k: v for k, v in st.items() if "pyr" in k})
    except Exception as e:
        print(n, "failed", e)
PY
