#!/bin/bash
# The CPU checker under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool: CPU build only).
# Builds oracle/ref_*.cpp with -fsanitize=address,undefined into /tmp (TAIL=cv flavour: OpenCV's ITERATIVE tail included) and runs the
# oracle tests against it through tests/_oracle.py's FLVIS_ORACLE_LIB override.  Round 6, fourth session: 108 tests, no report.
#   scripts/oracle_sanitizers.sh [pytest args...]        (default: every tests/test_oracle_*.py except the 100 s order comparison)
set -eu
cd "$(dirname "$0")/.."
mkdir -p /tmp/flvis_asan
( cd oracle && g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -DFLVIS_TAIL_CV \
    -shared -o /tmp/flvis_asan/libflvis_ref_asan.so ref_*.cpp -lpthread )
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) FLVIS_ORACLE_LIB=/tmp/flvis_asan/libflvis_ref_asan.so
if [ $# -gt 0 ]; then exec python -m pytest -q -x "$@"; fi
exec python -m pytest -q -x tests/test_oracle_*.py -k "not chunk_sums"
