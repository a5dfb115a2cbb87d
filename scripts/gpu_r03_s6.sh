#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_s6
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout 900 python "$R/bench.py" --pmc --steps 20 --warmup 10 < /dev/null > "$OUT/pmc.log" 2>&1; tail -c 400 "$OUT/pmc.log"; echo
cp "$R/gpurun_out/r03_kernel_pmc.json" "$R/gpurun_out/r03_lk_pmc.json" "$OUT/" 2>/dev/null
mkdir -p "$R/profiles"
timeout 300 python "$R/bench.py" < /dev/null > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 300 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r.get("leg_errors"))
for k in r["roofline"].get("kernels", []): print(json.dumps(k))
print(json.dumps(r["roofline"].get("lk_iterations")))
PY
