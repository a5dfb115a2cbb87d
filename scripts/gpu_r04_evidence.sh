#!/bin/bash
# round-4 evidence in one call: the whole GPU test suite, smoke(), then scripts/collect_profiles.sh r04
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r04
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 2400 python -m pytest tests -q -m gpu < /dev/null > "$OUT/r04_gpu_tests.log" 2>&1; tail -5 "$OUT/r04_gpu_tests.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
bash scripts/collect_profiles.sh r04
