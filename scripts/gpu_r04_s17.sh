#!/bin/bash
# round 4, session 17: the pyramid without LDS tiles (k_pyr_down_direct) against the tile kernels (FLVIS_PYR_TILES=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s17
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests/test_gpu_image.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "pyr or lk or frontend or cache or kitti or euroc or ingest or border" < /dev/null > "$OUT/gpu_tests.log" 2>&1; tail -6 "$OUT/gpu_tests.log"
B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run direct FLVIS_NOP=1
run tiles FLVIS_PYR_TILES=1
run direct2 FLVIS_NOP=1
python - "$OUT" <<'PY'
import json, sys
for n in ("b_direct", "b_tiles", "b_direct2"):
    try:
        r = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        st = r.get("stages_ms_per_step", {}) or {}
        print(n, r["value"], r["ms_per_step"], "chain p50", (r.get("latency_ms") or {}).get("gpu_frame_chain_p50"), "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"),
              {k: v for k, v in st.items() if "pyr" in k})
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o direct -- python "$R/bench.py" $B --steps 20 --warmup 5 < /dev/null > "$OUT/prof_direct.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pyr" in row["Name"] or "lk_track" in row["Name"]:
            print(row["Name"][:60], row["Calls"], row["AverageNs"], row["Percentage"])
PY
