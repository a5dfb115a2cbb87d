#!/bin/bash
# round 4, session 35: no prefetched value used before the chunk's arithmetic in the local map's Schur phase (the early wait gone): parity, phases, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s35
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "local_map" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -2 "$OUT/gpu_tests.log"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baprof.so timeout 200 python scripts/ba_prof.py 110 < /dev/null > "$OUT/ba_prof.txt" 2>&1
grep -v "amdgpu.ids\|ransac\|seven_point\|EPnP\|pose_lm" "$OUT/ba_prof.txt" | head -22
timeout 200 python bench.py --no-h2d --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_1.json" 2> "$OUT/b_1.err"
python - "$OUT" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1] + "/b_1.json").read().strip().splitlines()[-1])
st = r["stages_ms_per_step"]; l = r["latency_ms"]
print(r["value"], r["ms_per_step"], l["gpu_frame_chain_p50"], l["gpu_frame_chain_p99"], "ba launch", st.get("ba_worker(launch)"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"))
PY
