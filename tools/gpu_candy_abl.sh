#!/bin/bash
# Candy's fp16 convolution shapes (batch 8), normal build and the conv2d_mfma ablation builds (timing only, results wrong by construction)
cd $GRAFT_REPO_ROOT
SH="--shape 8,728,1288,3,32,9,1 --shape 8,180,320,128,128,3,1 --shape 8,360,640,64,32,3,1 --shape 8,720,1280,32,64,3,2 --shape 8,360,640,64,128,3,2"
echo "== normal"; timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-250
for n in 1 2 4 8 15; do
  echo "== abl $n"; SNNHIP_LIB_PATH=build/abl/libsnnhip_abl$n.so timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-90
done
