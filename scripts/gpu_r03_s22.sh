#!/bin/bash
# the with_h2d leg, repeated: is the rate bimodal, and where does the time go in the slow mode?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s22
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for i in 1 2 3 4 5; do
K=20; [ $((i % 2)) -eq 0 ] && K=60
FLVIS_BENCH_FRAMES=1 timeout 200 python "$R/bench.py" --gpus 1 --steps $K --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
python - "$OUT/b_$i.json" $K <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h = r["with_h2d"]
c = h.get("host_call_ms", [])
print("K", sys.argv[2], "value", r["value"], "h2d", h["value"], "loop_ms", h.get("loop_ms"), "total_ms", h.get("total_ms"), "calls p50 %.2f max %.2f first5 %s" % (sorted(c)[len(c)//2], max(c), c[:5]))
PY
done
