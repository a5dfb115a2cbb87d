#!/bin/bash
# round 4, session 31: the copy engines opened at the first host-image call: does the leg still have slow runs at the driver's arguments?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s31
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 600 python -m pytest tests -q -m gpu -k "host or cpp_caller or keyframe_msg" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -2 "$OUT/gpu_tests.log"
for i in 1 2 3 4 5 6; do
  FLVIS_BENCH_FRAMES=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    h = r.get("with_h2d") or {}
    c = h.get("host_call_ms") or []
    print(os.path.basename(f), r["value"], "h2d", h.get("value"), "max call after the warm-up", max(c[4:]) if len(c) > 4 else None, "first calls", [round(v, 1) for v in c[:5]])
PY
