#!/bin/bash
# round 4, session 16: local-map workgroups of 256 threads (one wave per SIMD at 253 VGPRs: half of the register file stays free for LK waves)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s16
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_ba256.so timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "local_map_parity" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -4 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run default FLVIS_NOP=1
run ba256 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_ba256.so
run ba256_lds96 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_ba256.so FLVIS_BA_LDS_KB=96
run ba256_lds64 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_ba256.so FLVIS_BA_LDS_KB=64
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_ba256", "b_ba256_lds96", "b_ba256_lds64"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              "ba", st.get("ba_worker(launch)"), "tail", ((r.get("latency_ms") or {}).get("timed_region_ms") or {}).get("local_map_tail_after_last_frame"))
    except Exception as e:
        print(n, "failed", e)
PY
