#!/bin/bash
# round 4, session 22: host images uploaded by a copy kernel (FLVIS_H2D_KERNEL=n workgroups) instead of the SDMA engine
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s22
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
FLVIS_H2D_KERNEL=16 timeout 900 python -m pytest tests -q -m gpu -k "host or cpp_caller or dataset or keyframe_msg" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -4 "$OUT/gpu_tests.log"
B="--cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run sdma FLVIS_NOP=1
run k8 FLVIS_H2D_KERNEL=8
run k16 FLVIS_H2D_KERNEL=16
run k32 FLVIS_H2D_KERNEL=32
run k128 FLVIS_H2D_KERNEL=128
run sdma2 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys
for n in ("b_sdma", "b_k8", "b_k16", "b_k32", "b_k128", "b_sdma2"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        print(n, r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"), "h2d", (r.get("with_h2d") or {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
