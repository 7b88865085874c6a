#!/bin/bash
# Round 6: conv2d_ksplit ablation builds (tools/exp_one.sh conv2d_ksplit.hip tag:-DSNNHIP_KS_ABL=n) on the three ResNet-18 stride-2 layers, default geometry.
cd "$(dirname "$0")/.."
SH="--shape 32,56,56,64,128,3,2 --shape 32,28,28,128,256,3,2 --shape 32,14,14,256,512,3,2 --only adhoc --reps 200"
for t in "" ${1:-ksw ksa kswa ksr ksall}; do
  if [ -n "$t" ]; then export SNNHIP_LIB_PATH=$PWD/build/abl/libsnnhip_$t.so; fi
  python tools/bench_layers.py $SH 2>/dev/null | awk -v t="${t:-product}" '{printf "%-8s %s\n", t, $0}' | cut -c1-130
done
