#!/bin/bash
# Round 6: conv2d_ksplit geometry sweep on ResNet-18's three 3x3 stride-2 layers at batch 32 (us per launch; first line = the split-K + reduce kernel it replaces).
#   usage (GPU box, via tools/gpu.sh): sh:r6_ks.sh[:"geom geom ..."]
cd "$(dirname "$0")/.."
SH="--shape 32,56,56,64,128,3,2 --shape 32,28,28,128,256,3,2 --shape 32,14,14,256,512,3,2 --only adhoc --reps 200"
run() { python tools/bench_layers.py $SH 2>/dev/null | awk -v t="$1" '{printf "%-10s %s\n", t, $0}' | cut -c1-250; }
SNNHIP_CONV_KSPLIT=0 run splitk
run default
for g in ${1:-1,4,2 1,8,2 2,4,2 2,8,2 2,16,2 1,4,3 1,8,3 2,4,3 2,8,3 2,2,2}; do SNNHIP_KSPLIT=$g run "$g"; done
