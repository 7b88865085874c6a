#!/bin/bash
# Round 6: conv2d_wino ablation builds (tools/ablate_wino.sh) on the ResNet-18 body layers at batch 32 (us per launch incl. the reduce pass of the split-K layers).
cd "$(dirname "$0")/.."
for t in "" ${1:-255 132 64 48 8 1}; do
  if [ -n "$t" ]; then export SNNHIP_LIB_PATH=$PWD/build/abl/libsnnhip_wabl$t.so; fi
  python tools/bench_layers.py --only "resnet l" --reps 200 2>/dev/null | grep "3x3 " | grep -v s2 | awk -v t="${t:-product}" '{printf "%-8s %s\n", t, $0}' | cut -c1-120
done
