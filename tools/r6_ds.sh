#!/bin/bash
# ResNet-18's three 1x1 stride-2 downsample layers (batch 32) over the column width (NT x 32 channels per block) and waves per block of conv1x1_stream.
cd "$(dirname "$0")/.."
S="--shape 32,56,56,64,128,1,2 --shape 32,28,28,128,256,1,2 --shape 32,14,14,256,512,1,2"
for nt in 3 2 1; do for wv in 4 8; do
  echo "== NT<=$nt waves=$wv"
  SNNHIP_CONV_1X1_NT=$nt SNNHIP_CONV_1X1_WAVES=$wv python tools/bench_layers.py --only=adhoc $S --reps 200 2>&1 | grep -v "^#" | tail -4
done; done
echo "== general kernel (SNNHIP_CONV_1X1=0)"
SNNHIP_CONV_1X1=0 python tools/bench_layers.py --only=adhoc $S --reps 200 2>&1 | tail -4
