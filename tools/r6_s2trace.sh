#!/bin/bash
cd "$(dirname "$0")/.."
SNNHIP_LIB_PATH=$PWD/build/abl/libsnnhip_s2trace.so python tools/bench_layers.py --fp16 --only=adhoc --shape 16,720,1280,32,64,3,2 --shape 16,360,640,64,128,3,2 --reps 2 2>&1 | grep -i "trace\|adhoc" | head -24
