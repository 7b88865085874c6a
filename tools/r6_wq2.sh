#!/bin/bash
# conv2d_wino residency experiment (round 6): the ResNet-18 body layers at batch 32 (and a few batches around it) with two blocks per CU (product) and with
# extra LDS that leaves ONE block per CU (SNNHIP_WINO_LDS_PAD).   tools/gpu.sh <tag> sh:r6_wq2.sh
cd "$GRAFT_REPO_ROOT"
args=""
for n in 20 31 32 40 64; do args="$args --shape $n,56,56,64,64,3,1"; done
for n in 32; do args="$args --shape $n,28,28,128,128,3,1 --shape $n,14,14,256,256,3,1 --shape $n,7,7,512,512,3,1"; done
for pad in 0 32768 0 32768; do
  echo "== SNNHIP_WINO_LDS_PAD=$pad"
  SNNHIP_WINO_LDS_PAD=$pad python tools/bench_layers.py --only adhoc --reps 100 $args 2>/dev/null | cut -c1-175
done
