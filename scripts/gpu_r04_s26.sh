#!/bin/bash
# round 4, session 26: Schur slices in proportion to the landmarks of a pose pair: local-map parity, phase profile, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s26
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_ba.py -q -m gpu -k "local_map or config or ba" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -5 "$OUT/gpu_tests.log"
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_baprof.so timeout 300 python scripts/ba_prof.py 110 < /dev/null > "$OUT/ba_prof.txt" 2>&1
grep -v "amdgpu.ids" "$OUT/ba_prof.txt" | head -20
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
for i in 1 2 3 4; do
  timeout 300 python bench.py $B < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        st = r.get("stages_ms_per_step", {}) or {}
        print(os.path.basename(f), r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "ba launch", st.get("ba_worker(launch)"),
              "tail", (l.get("timed_region_ms") or {}).get("local_map_tail_after_last_frame"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"))
    except Exception as e:
        print(f, "failed", e)
PY
