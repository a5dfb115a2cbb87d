#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s8
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -x -k "run_out" < /dev/null > "$OUT/t1.log" 2>&1; tail -5 "$OUT/t1.log"
cd /tmp || exit 1
FLVIS_BENCH_FRAMES=1 timeout 300 python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 < /dev/null > "$OUT/bench_h2d.json" 2> "$OUT/bench_h2d.err"
python - "$OUT/bench_h2d.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r.get("with_h2d"))
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/s8_tl -o t -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/tl.log" 2>&1
K=$(find /tmp/s8_tl -name "*kernel_trace.csv" | head -1); M=$(find /tmp/s8_tl -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" > "$OUT/h2d_timeline.txt" <<'PY'
import csv, sys
ev = []
for x in csv.DictReader(open(sys.argv[1])):
    n = x["Kernel_Name"]
    if "k_frame_head" in n or "k_frame_end" in n or "k_store_progress" in n:
        ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n.split("(")[0].split("::")[-1]))
for x in csv.DictReader(open(sys.argv[2])):
    b = int(x.get("Size", x.get("size", 0)) or 0) if ("Size" in x or "size" in x) else 0
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), "COPY %s %d B" % (x.get("Direction", x.get("Name", "?")), b)))
ev.sort()
ev = ev[-260:]
t0 = ev[0][0]
for a, b, n in ev:
    if n.startswith("COPY") and "B" in n and int(n.split()[-2]) < 1000000: continue
    print("%-40s start %10.1f us  dur %8.1f us" % (n, (a - t0) / 1e3, (b - a) / 1e3))
PY
tail -70 "$OUT/h2d_timeline.txt"
