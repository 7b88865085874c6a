#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "base"; timeout 300 python tools/bench_layers.py --only "3x3" 2>/dev/null | grep -i "l1 3x3\|unet 3x3 128" | cut -c1-75
for n in 1 4 64 128; do
echo "abl $n"; SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_wabl$n.so timeout 300 python tools/bench_layers.py --only "3x3" 2>/dev/null | grep -i "l1 3x3\|unet 3x3 128" | cut -c1-75
done
