#!/bin/bash
# round 4, session 30: the host-image leg with more than one frame of host lead (FLVIS_H2D_LEAD), after the local map's launches were shortened
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_s30
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="--cpu-frames 0 --cpu-mt-frames 0"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $B < /dev/null > "$OUT/b_$n.json" 2> "$OUT/b_$n.err"; }
run l1_a FLVIS_NOP=1
run l2_a FLVIS_H2D_LEAD=2
run l3_a FLVIS_H2D_LEAD=3
run l1_b FLVIS_NOP=1
run l2_b FLVIS_H2D_LEAD=2
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), r["value"], "h2d", (r.get("with_h2d") or {}).get("value"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
