#!/bin/bash
# Round 6: ResNet-18's three 1x1 stride-2 downsample layers (batch 32): conv1x1_stream (product) against conv2d_ksplit forced on them, per geometry.
cd "$(dirname "$0")/.."
SH="--shape 32,56,56,64,128,1,2 --shape 32,28,28,128,256,1,2 --shape 32,14,14,256,512,1,2 --only adhoc --reps 300"
run() { python tools/bench_layers.py $SH $2 2>/dev/null | awk -v t="$1" '{printf "%-10s %s\n", t, $0}' | cut -c1-200; }
run stream
for g in ${1:-1,1,2 1,2,2 2,1,2 2,2,2 1,4,2}; do SNNHIP_KSPLIT=$g run "$g" "--force ksplit"; done
