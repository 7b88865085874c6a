#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_wide3}; mkdir -p "$O"
timeout 420 python -m pytest tests/test_conv_wide_gpu.py -m gpu -q -x --timeout 120 -k "wide" > "$O/t_wide.txt" 2>&1; rc=$?; echo "wide tests rc=$rc"; tail -5 "$O/t_wide.txt" | cut -c1-220
if [ $rc -ne 0 ]; then tail -40 "$O/t_wide.txt" | cut -c1-200; exit 0; fi
for lib in "" prio1; do
  [ -n "$lib" ] && export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_$lib.so
  echo "#### lib=$lib"
  timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --layer-table 0 > "$O/bench_c5_$lib.json" 2> "$O/bench_c5_$lib.err" || tail -3 "$O/bench_c5_$lib.err"
  python tools/bench_digest.py "$O/bench_c5_$lib.json" | head -3
done
unset SNNHIP_LIB_PATH
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_fp16_gpu.py tests/test_guard_gpu.py -m gpu -q --timeout 600 -k "c5 or candy or guard or style" > "$O/t_c5.txt" 2>&1; echo "c5 tests rc=$?"; tail -4 "$O/t_c5.txt" | cut -c1-220
export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_wptrace.so
timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --no-parity --layer-table 0 --event-launches 0 --preheat-ms 0 --steps 1 --warmup 0 --repeats 1 > "$O/c5trace.txt" 2>&1
grep wptrace "$O/c5trace.txt" | grep "blk 300 tid 0" | tail -4
