#!/bin/bash
# The AQL packets of the host-image leg as the HIP runtime logs them (AMD_LOG_LEVEL=4): per kernel its queue, the header's barrier bit and
# acquire / release fence scopes, its completion signal; the barrier packets with their dependency signals.  Filtered on the fly (the raw
# log is hundreds of MB); the tail of the filtered log = the last calls of the leg (the pose check is skipped: FLVIS_BENCH_H2D_NOCHECK).
#   scripts/h2d_aql_log.sh OUTFILE [ENV=V ...]
OUT=${1:?out}; shift
env "$@" AMD_LOG_LEVEL=4 FLVIS_BENCH_H2D_NOCHECK=1 timeout 900 python bench.py --steps 4 --warmup 2 --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue \
  2>&1 >/dev/null | grep -E "ShaderName|Dispatch Header|Header =|Barrier|hipStreamWaitEvent|hipEventRecord|hipMemcpyAsync|hipLaunchKernel|Copy|copy|signal" \
  | cut -c1-420 | tail -n 9000 > "$OUT"
wc -l "$OUT"
