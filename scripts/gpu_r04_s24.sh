#!/bin/bash
# round 4, session 24: per-frame chain / stage / host-call times of default runs, to see what the occasional 3-8 ms frame chains consist of
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s24
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
for i in 1 2 3 4 5; do
  FLVIS_BENCH_FRAMES=1 timeout 300 python bench.py $B < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
done
for i in 6 7; do
  FLVIS_HOST_LEAD=1 FLVIS_BENCH_FRAMES=1 timeout 300 python bench.py $B < /dev/null > "$OUT/b_$i.json" 2> "$OUT/b_$i.err"
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); continue
    l = r["latency_ms"]
    ch, hc, fs = l["frame_chain_ms"], l["host_call_ms"], l["frame_stage_ms"]
    print(os.path.basename(f), r["value"], r["ms_per_step"], "p50/p99", l["gpu_frame_chain_p50"], l["gpu_frame_chain_p99"], "host wait", l["host_in_image_feed_ms"])
    for i, c in enumerate(ch):
        if c > 1.5:
            big = {k: v[i] for k, v in fs.items() if i < len(v) and v[i] > 0.35 and "frame(" not in k}
            tot = sum(v[i] for k, v in fs.items() if i < len(v) and "frame(" not in k and "ba_worker" not in k and "pyr_down(left)" not in k and "gftt" not in k)
            print("   frame", i, "chain", c, "host calls around", hc[max(0, i - 2):i + 3], "stages > 0.35 ms:", big, "sum of main-stream stages", round(tot, 3))
PY
