#!/bin/bash
# round 4, session 29: the host-image leg's per-call times at the driver's arguments (where do its occasional slow runs come from?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s29
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for i in 1 2 3 4; do
  FLVIS_BENCH_FRAMES=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    h = r.get("with_h2d") or {}
    c = h.get("host_call_ms") or []
    print(os.path.basename(f), r["value"], "h2d", h.get("value"), "steps", h.get("steps"), "loop", h.get("loop_ms"), "total", h.get("total_ms"))
    print("   calls:", " ".join("%.1f" % v for v in c))
PY
