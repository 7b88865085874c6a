#!/bin/bash
# Round 6: MobileNetV2's widening pointwise layers at batch 256 (conv1x1_stream_kernel's two launches in c4) against conv2d_ksplit forced on them.
cd "$(dirname "$0")/.."
SH="--shape 256,7,7,320,1280,1,1 --shape 256,7,7,160,960,1,1 --shape 256,14,14,96,576,1,1 --shape 256,14,14,64,384,1,1 --only adhoc --reps 200"
run() { python tools/bench_layers.py $SH $2 2>/dev/null | awk -v t="$1" '{printf "%-10s %s\n", t, $0}' | cut -c1-210; }
run product
for g in ${1:-2,1,2 2,2,2 2,4,2 1,2,2 1,4,2}; do SNNHIP_KSPLIT=$g run "$g" "--force ksplit"; done
