#!/bin/bash
# round 5: phase sums of the persistent wide kernel (experiment build -DSNNHIP_WIDEP_TRACE: one block sums the s_memtime spans of its phases over its tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_trace}; mkdir -p "$O"
export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_wptrace.so
timeout 300 python tools/bench_layers.py --fp16 --only adhoc --shape 16,183,323,128,128,3,1 --reps 3 > "$O/layer.txt" 2>&1
grep wptrace "$O/layer.txt" | tail -8; grep adhoc "$O/layer.txt" | cut -c1-100
timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --no-parity --layer-table 0 --event-launches 0 --preheat-ms 0 --steps 1 --warmup 0 --repeats 1 > "$O/c5.txt" 2>&1
echo "c5 graph (all ten layers of the last micro-batch):"; grep wptrace "$O/c5.txt" | tail -40
