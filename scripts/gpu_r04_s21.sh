#!/bin/bash
# round 4, session 21: k_frame_head + k_track_prepare in one launch; the three-level walking pyramid (FLVIS_PYR_PLAN=3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s21
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 600 python -m pytest tests/test_gpu_image.py -q -m gpu -k "pyr" < /dev/null > "$OUT/gpu_tests_pyr.log" 2>&1; tail -3 "$OUT/gpu_tests_pyr.log"
FLVIS_PYR_PLAN=3 timeout 600 python -m pytest tests/test_gpu_image.py -q -m gpu -k "pyr" < /dev/null > "$OUT/gpu_tests_pyr3.log" 2>&1; tail -3 "$OUT/gpu_tests_pyr3.log"
FLVIS_PYR_PLAN=21 timeout 600 python -m pytest tests/test_gpu_image.py -q -m gpu -k "pyr" < /dev/null > "$OUT/gpu_tests_pyr21.log" 2>&1; tail -3 "$OUT/gpu_tests_pyr21.log"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "frontend or imu or kitti or euroc or cpp_caller or config or feedback" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -4 "$OUT/gpu_tests.log"
cd /tmp
one() { n=$1; shift; rm -rf /tmp/s21_$n; env "$@" PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s21_$n -o b -- python "$R/scripts/pyr_bench.py" 64 40 ${ING:-1} < /dev/null > "$OUT/$n.log" 2>&1
  S=$(find /tmp/s21_$n -name "*kernel_stats.csv" | head -1); echo "== $n"; [ -n "$S" ] && grep -E "pyr" "$S" | sed 's/flvis::(anonymous namespace):://g; s/(WalkArgs)//' | cut -d, -f1-5 | tee "$OUT/$n.csv"; }
one walk12 FLVIS_NOP=1
one walk3 FLVIS_PYR_PLAN=3
one walk3_b1 FLVIS_PYR_PLAN=3 FLVIS_PYR_BAND3=1
one walk3_b4 FLVIS_PYR_PLAN=3 FLVIS_PYR_BAND3=4
ING=0 one walk3_noingest FLVIS_PYR_PLAN=3
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run default FLVIS_NOP=1
run twolaunch FLVIS_HEAD_PREPARE=0
run plan3 FLVIS_PYR_PLAN=3
run plan3_b4 FLVIS_PYR_PLAN=3 FLVIS_PYR_BAND3=4
run default2 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys
for n in ("b_default", "b_twolaunch", "b_plan3", "b_plan3_b4", "b_default2"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        l = r.get("latency_ms") or {}
        print(n, r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              {k: v for k, v in st.items() if "pyr" in k or "head" in k or "prepare" in k})
    except Exception as e:
        print(n, "failed", e)
PY
