#!/bin/bash
# same-box A/B of conv2d_wide_f16 experiment / ablation builds (tools/exp_wide.sh) on Candy's layer shapes (batch 16)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec > >(tee gpurun_out/wide_exp.txt) 2>&1
echo "# conv2d_wide_f16 ablation / experiment builds (tools/exp_wide.sh), same box, tools/bench_layers.py; ABL bits: 1 no weight loads, 2 no LDS operand reads, 4 no activation DMA, 8 no output stores (results wrong by construction: timing only)"
SH="--shape 16,816,1376,64,32,3,1 --shape 16,408,688,128,64,3,1 --shape 16,183,323,128,128,3,1"
echo "== normal"; timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-200
for t in "$@"; do
  echo "== $t"; SNNHIP_LIB_PATH=build/abl/libsnnhip_$t.so timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-100
done
echo "== normal"; timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-100
