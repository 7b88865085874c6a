#!/bin/bash
# Round 6: the classifier heads as 1x1 "convolutions" on conv2d_ksplit (32 x 512 -> 1024, 256 x 1280 -> 1024; dense_mfma_kernel: 9.0 / 17.8 us in the graphs)
cd "$(dirname "$0")/.."
SH="--shape 32,1,1,512,1024,1,1 --shape 256,1,1,1280,1024,1,1 --only adhoc --reps 300"
for g in ${1:-1,8,2 1,4,2 1,2,2 2,8,2 2,4,2}; do SNNHIP_KSPLIT=$g python tools/bench_layers.py $SH --force ksplit 2>/dev/null | awk -v t="$g" '{printf "%-10s %s\n", t, $0}' | cut -c1-200; done
