#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_wide2}; mkdir -p "$O"
timeout 420 python -m pytest tests/test_conv_wide_gpu.py -m gpu -q -x --timeout 120 -k "wide" > "$O/t_wide.txt" 2>&1; rc=$?; echo "wide tests rc=$rc"; tail -5 "$O/t_wide.txt" | cut -c1-220
if [ $rc -ne 0 ]; then tail -40 "$O/t_wide.txt" | cut -c1-200; exit 0; fi
tools/ab_wide.sh SNNHIP_WIDE_PERSIST=0 SNNHIP_WIDE_PERSIST=1  2>&1 | grep -v "128->64" | tee "$O/ab.txt"
for spec in SNNHIP_WIDE_PERSIST=0 SNNHIP_WIDE_PERSIST=1; do
  env $spec timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --layer-table 0 > "$O/bench_c5_$spec.json" 2> "$O/bench_c5_$spec.err" || tail -3 "$O/bench_c5_$spec.err"
  python tools/bench_digest.py "$O/bench_c5_$spec.json" | head -4
done
export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_wptrace.so
timeout 300 python tools/bench_layers.py --fp16 --only adhoc --shape 15,183,323,128,128,3,1 --reps 1 > "$O/census.txt" 2>&1
grep -c wpblk "$O/census.txt"
timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --no-parity --layer-table 0 --event-launches 0 --preheat-ms 0 --steps 1 --warmup 0 --repeats 1 > "$O/c5trace.txt" 2>&1
grep wptrace "$O/c5trace.txt" | grep "blk 300 tid 0" | tail -4
