#!/bin/bash
# round 4, session 27: the local-map solver's phases as calls or inlined; time of a keyframe's bookkeeping
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s27
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for v in baprof baprof_inl; do
  FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_$v.so timeout 300 python scripts/ba_prof.py 110 < /dev/null > "$OUT/$v.txt" 2>&1
  echo "== $v"; grep -v "amdgpu.ids" "$OUT/$v.txt" | head -22
done
FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_bainl.so timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "local_map" < /dev/null > "$OUT/gpu_tests_inl.log" 2>&1; tail -3 "$OUT/gpu_tests_inl.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run inl1 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_bainl.so
run def1 FLVIS_NOP=1
run inl2 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_bainl.so
run def2 FLVIS_NOP=1
run inl3 FLVIS_LIB_PATH=$R/build_variants/libflvis_hip_bainl.so
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
