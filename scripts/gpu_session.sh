#!/bin/bash
# One parameterised GPU session (run through gpurun from the repo root); replaces the per-session scripts of rounds 3-4:
#   scripts/gpu_session.sh TAG STEP [STEP ...]
# Everything a step writes goes to gpurun_out/TAG/.  Steps, executed in order, each under its own timeout:
#   tests=<pytest args>          python -m pytest -q -m gpu <args> (eval'ed: -k 'a or b' may be quoted)  -> tests_<n>.log
#   smoke                        __graft_entry__.smoke()                                -> smoke.log
#   env=<K=V> / unset=<K>        an environment variable for the steps that follow
#   sh=<command line>            bash -c <command line>                                 -> sh_<n>.log
#   lib=<path|->                 FLVIS_LIB_PATH for the steps that follow (a build of scripts/build_variant.sh; "-": the in-tree library)
#   bench=<name>[,ENV=V...][,--flag...]  bench.py without the CPU / host-image legs (entries starting with -- go to bench.py) -> b_<name>.json
#   benchh2d=<name>[,ENV=V...]   bench.py with the host-image leg, without the CPU legs -> b_<name>.json
#   benchfull=<name>[,ENV=V...]  bench.py with every leg (the driver's arguments)       -> b_<name>.json
#   baprof[=<frames>]            scripts/ba_prof.py (needs a -DFLVIS_BA_PROF variant selected by lib=)  -> ba_prof_<n>.txt
#   py=<script and args>         python <script and args>                               -> py_<n>.log
#   trace=<name>[,ENV=V...]      rocprofv3 --kernel-trace --stats of bench.py (60 steps) -> <name>_kernel_summary.md, <name>_timeline.txt
#   traceh2d=<name>[,ENV=V...]   rocprofv3 kernel + memory-copy trace of bench.py with its host-image leg -> <name>_h2d_kernels.txt (per-kernel
#                                durations, resident frames vs host-image frames), <name>_h2d_timeline.txt
#   evidence=<rNN>               the whole GPU suite + smoke + scripts/collect_profiles.sh rNN (the round's last session)
# After the steps a table of every b_*.json of the session is printed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:?tag}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
n=0
split_env() {  # name[,ENV=V...][,--bench-flag...]: entries that start with "--" are passed to bench.py, the others to env
  IFS=',' read -r -a parts <<< "$1"; name=${parts[0]}; envs=(); xargs=()
  for q in "${parts[@]:1}"; do case "$q" in --*) xargs+=("$q") ;; *) envs+=("$q") ;; esac; done
}
for step in "$@"; do
  n=$((n + 1))
  key=${step%%=*}; val=""; [ "$key" != "$step" ] && val=${step#*=}
  case "$key" in
    tests) eval "timeout 2400 python -m pytest -q -m gpu $val" < /dev/null > "$OUT/tests_$n.log" 2>&1; tail -4 "$OUT/tests_$n.log" ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    lib) if [ "$val" = "-" ]; then unset FLVIS_LIB_PATH; else export FLVIS_LIB_PATH="$R/$val"; fi ;;
    env) export "$val" ;;
    unset) unset "$val" ;;
    sh) timeout 1200 bash -c "$val" < /dev/null > "$OUT/sh_$n.log" 2>&1; tail -15 "$OUT/sh_$n.log" ;;
    bench|benchh2d|benchfull)
      split_env "$val"
      case "$key" in
        bench) B="--no-h2d --cpu-frames 0 --cpu-mt-frames 0" ;;
        benchh2d) B="--cpu-frames 0 --cpu-mt-frames 0" ;;
        *) B="--gpus 1 --steps 20 --warmup 5" ;;
      esac
      env "${envs[@]}" timeout 420 python bench.py $B "${xargs[@]}" < /dev/null > "$OUT/b_$name.json" 2> "$OUT/b_$name.err" || tail -3 "$OUT/b_$name.err" ;;
    baprof) timeout 600 python scripts/ba_prof.py ${val:-110} < /dev/null > "$OUT/ba_prof_$n.txt" 2>&1; cat "$OUT/ba_prof_$n.txt" ;;
    py) timeout 900 python $val < /dev/null > "$OUT/py_$n.log" 2>&1; tail -30 "$OUT/py_$n.log" ;;
    trace)
      split_env "$val"
      rm -rf /tmp/gs_trace
      (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs_trace -o b -- python "$R/bench.py" --steps 60 --warmup 10 \
        --cpu-frames 0 --cpu-mt-frames 0 --no-h2d --no-epilogue < /dev/null > "$OUT/trace_$name.log" 2>&1)
      T=$(find /tmp/gs_trace -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python scripts/rocprof_summary.py "$T" 60 "$OUT/${name}_kernel_summary.md" < /dev/null > /dev/null
      [ -n "$T" ] && python scripts/timeline.py "$T" < /dev/null > "$OUT/${name}_timeline.txt"
      [ -n "$T" ] && python scripts/ba_launches.py "$T" < /dev/null > "$OUT/${name}_ba_launches.txt" && tail -12 "$OUT/${name}_ba_launches.txt"
      [ -f "$OUT/${name}_kernel_summary.md" ] && head -30 "$OUT/${name}_kernel_summary.md" ;;
    traceh2d)
      split_env "$val"
      rm -rf /tmp/gs_trace
      (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gs_trace -o b -- python "$R/bench.py" --steps 40 --warmup 10 \
        --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/trace_$name.log" 2>&1)
      T=$(find /tmp/gs_trace -name "*kernel_trace.csv" | head -1); M=$(find /tmp/gs_trace -name "*memory_copy_trace.csv" | head -1)
      [ -n "$T" ] && [ -n "$M" ] && python scripts/h2d_kernel_compare.py "$T" "$M" < /dev/null > "$OUT/${name}_h2d_kernels.txt" 2>&1 && cat "$OUT/${name}_h2d_kernels.txt"
      [ -n "$T" ] && [ -n "$M" ] && python scripts/h2d_full_timeline.py "$T" "$M" 2 < /dev/null > "$OUT/${name}_h2d_timeline.txt" 2>&1 ;;
    traceapi)
      split_env "$val"
      rm -rf /tmp/gs_trace
      (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gs_trace -o b -- python "$R/bench.py" --steps 20 --warmup 5 \
        --cpu-frames 0 --cpu-mt-frames 0 --no-epilogue < /dev/null > "$OUT/trace_$name.log" 2>&1)
      A=$(find /tmp/gs_trace -name "*hip_api_trace.csv" | head -1); T=$(find /tmp/gs_trace -name "*kernel_trace.csv" | head -1); M=$(find /tmp/gs_trace -name "*memory_copy_trace.csv" | head -1)
      [ -n "$A" ] && python scripts/h2d_api_timeline.py "$A" "$T" "$M" < /dev/null > "$OUT/${name}_h2d_api.txt" 2>&1; head -80 "$OUT/${name}_h2d_api.txt" ;;
    evidence)
      timeout 2700 python -m pytest tests -q -m gpu < /dev/null > "$OUT/${val}_gpu_tests.log" 2>&1; tail -5 "$OUT/${val}_gpu_tests.log"
      timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
      bash scripts/collect_profiles.sh "$val" ;;
    *) echo "unknown step $step" ;;
  esac
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/b_*.json"), key=os.path.getmtime):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        l = r.get("latency_ms") or {}
        st = r.get("stages_ms_per_step", {}) or {}
        h = r.get("with_h2d") or {}
        print(os.path.basename(f), r["value"], r["ms_per_step"], "chain p50/p99", l.get("gpu_frame_chain_p50"), l.get("gpu_frame_chain_p99"),
              "lk", st.get("lk_track(temporal)"), st.get("lk_track(stereo)"), "h2d", h.get("value") or h.get("value_unchecked"), h.get("poses_bit_identical"),
              "up GB/s", (h.get("link") or {}).get("upload_GBs"), (h.get("link") or {}).get("upload_alone_GBs"), "h2d chain", h.get("gpu_frame_chain_p50_ms"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
