#!/bin/bash
# The one script that runs on the GPU box (via gpurun): a sequence of actions, results under gpurun_out/<tag>/.
#   usage: gpurun --timeout S -- 'tools/gpu.sh <tag> <action> [<action> ...]'
#   actions:
#     tests[:<pytest -k expression>]   the -m gpu suite (or the tests matching the expression)            -> pytest[_N].txt
#     bench[:c1,c2,...]                one bench.py line per config (default: all five), each with --also none -> bench_<c>.json / .err
#     benchall                         the driver's command: python bench.py (c2 + the other four configs in `configs`) -> bench_all.json
#     benchx:<c>:<extra bench args>    one bench line with extra arguments ('+' stands for a blank)        -> benchx_<n>.json
#     prof:<c>                         rocprofv3 kernel trace + PMC passes of that config's bench command -> <tag>_<c>/summary.md
#     livecheck:<c>                    bench.py with its live launch trace under rocprofv3 --kernel-trace, same process -> live_vs_rocprof_<c>.json
#     profx:<name>:<command>           the same passes around an arbitrary command ('+' stands for a blank)   -> <tag>_<name>/summary.md
#     layers[:fp16]                    tools/bench_layers.py                                               -> layers[_fp16].txt
#     py:<script>[:args]               python tools/<script> args ('+' stands for a blank)                 -> py_<n>.txt
#     sh:<script>[:args]               bash tools/<script> args ('+' stands for a blank)                   -> sh_<n>.txt
#     lib:<tag>                        load build/abl/libsnnhip_<tag>.so in the following actions ('lib:' = the product library again)
#     env:NAME=VALUE                   export for the following actions
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-run}; shift
O=gpurun_out/$TAG
mkdir -p "$O"
export SNN_GIT_HEAD=$(cat .git_head 2>/dev/null)
n=0
for act in "$@"; do
  n=$((n + 1))
  kind=${act%%:*}; arg=""; [ "$kind" != "$act" ] && arg=${act#*:}
  case $kind in
    env) export "$arg" ;;
    tests)
      if [ -n "$arg" ]; then timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -k "${arg//+/ }" > "$O/pytest_$n.txt" 2>&1; tail -5 "$O/pytest_$n.txt"
      else timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > "$O/pytest.txt" 2>&1; tail -6 "$O/pytest.txt"; fi ;;
    bench)
      for c in $(echo "${arg:-c2,c1,c3,c4,c5}" | tr ',' ' '); do
        timeout 900 python bench.py --config "$c" --also none --detail-out "$O/bench_$c.detail.json" > "$O/bench_$c.json" 2> "$O/bench_$c.err" || { echo "bench $c rc=$?"; tail -3 "$O/bench_$c.err"; }
      done
      python tools/bench_digest.py "$O" ;;
    benchall)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$O/bench_all.detail.json" > "$O/bench_all.json" 2> "$O/bench_all.err" || { echo "benchall rc=$?"; tail -3 "$O/bench_all.err"; }
      python tools/bench_digest.py "$O/bench_all.json" ;;
    benchx)
      c=${arg%%:*}; extra=${arg#*:}
      timeout 900 python bench.py --config "$c" ${extra//+/ } --detail-out "$O/benchx_$n.detail.json" > "$O/benchx_$n.json" 2> "$O/benchx_$n.err" || { echo "benchx $arg rc=$?"; tail -3 "$O/benchx_$n.err"; }
      python tools/bench_digest.py "$O/benchx_$n.json" ;;
    prof)
      steps=$(python -c "print({'c1':'--steps 200 --warmup 20','c2':'--steps 100 --warmup 10','c3':'--steps 10 --warmup 2','c4':'--steps 4 --warmup 1','c5':'--steps 2 --warmup 1'}['$arg'])")
      PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --config $arg --also none --no-cpu-baseline --no-parity --layer-table 0 --event-launches 0 --preheat-ms 20 --repeats 1 $steps" \
        timeout 1500 tools/profile_gpu.sh "${TAG}_$arg" > "$O/profile_$arg.log" 2>&1
      grep "derived" -A30 "$O/profile_$arg.log" | cut -c1-240 | head -34 ;;
    livecheck) # livecheck:<c>: bench.py (its own launch trace ON) under rocprofv3 --kernel-trace in one process; tools/live_vs_rocprof.py compares the two on the same launches
      mkdir -p "$O/live_$arg"
      ( cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/live_$arg" -- python "$GRAFT_REPO_ROOT/bench.py" --config "$arg" --also none --no-cpu-baseline --layer-table 0 --detail-out "$GRAFT_REPO_ROOT/$O/live_$arg/detail.json" > "$GRAFT_REPO_ROOT/$O/live_$arg/line.json" 2> "$GRAFT_REPO_ROOT/$O/live_$arg/err.txt" )
      python tools/live_vs_rocprof.py "$O/live_$arg" | tee "$O/live_vs_rocprof_$arg.json" ;;
    profx) # profx:<name>:<command, '+' for blanks>: kernel trace + PMC passes of an arbitrary command (e.g. one layer through tools/bench_layers.py)
      nm=${arg%%:*}; cmd=${arg#*:}
      PROF_CMD="${cmd//+/ }" timeout 1500 tools/profile_gpu.sh "${TAG}_$nm" > "$O/profile_$nm.log" 2>&1
      grep "derived" -A12 "$O/profile_$nm.log" | cut -c1-600 | head -16 ;;
    layers)
      timeout 600 python tools/bench_layers.py ${arg:+--$arg} > "$O/layers${arg:+_$arg}.txt" 2>/dev/null; tail -40 "$O/layers${arg:+_$arg}.txt" ;;
    lib) export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$arg.so"; [ -z "$arg" ] && unset SNNHIP_LIB_PATH ;;   # lib:<tag> = an experiment build (tools/exp_one.sh), lib: = back to the product library
    py)
      s=${arg%%:*}; a=""; [ "$s" != "$arg" ] && a=${arg#*:}
      timeout 1200 python "tools/$s" ${a//+/ } > "$O/py_$n.txt" 2>&1; tail -60 "$O/py_$n.txt" ;;
    sh)
      s=${arg%%:*}; a=""; [ "$s" != "$arg" ] && a=${arg#*:}
      timeout 1800 bash "tools/$s" ${a//+/ } > "$O/sh_$n.txt" 2>&1; echo "sh $s rc=$?"; tail -25 "$O/sh_$n.txt" ;;
    *) echo "unknown action $act" ;;
  esac
done
