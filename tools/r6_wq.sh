#!/bin/bash
# conv2d_wino block quantisation (round 6): the 56x56 x 64 -> 64 layer of ResNet-18 at batches around 32 (block tiles x 2 channel blocks: 24.5 per image;
# 512 block slots on the chip): us per launch and us per image.   tools/gpu.sh <tag> sh:r6_wq.sh
cd "$GRAFT_REPO_ROOT"
args=""
for n in 20 21 24 28 30 31 32 33 36 40 41 42 48 62 63 64; do args="$args --shape $n,56,56,64,64,3,1"; done
for n in 31 32 62 63 64; do args="$args --shape $n,28,28,128,128,3,1 --shape $n,14,14,256,256,3,1 --shape $n,7,7,512,512,3,1"; done
python tools/bench_layers.py --only adhoc --reps 100 $args 2>/dev/null | cut -c1-200
