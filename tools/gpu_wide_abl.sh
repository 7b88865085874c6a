#!/bin/bash
cd $GRAFT_REPO_ROOT
SH="--shape 8,180,320,128,128,3,1 --shape 8,360,640,64,32,3,1 --shape 8,360,640,128,64,3,1"
echo "== normal"; timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-250
for n in 1 2 4 8 15; do
  echo "== abl $n"; SNNHIP_LIB_PATH=build/abl/libsnnhip_wide$n.so timeout 300 python tools/bench_layers.py --fp16 --only adhoc $SH 2>/dev/null | cut -c1-90
done
