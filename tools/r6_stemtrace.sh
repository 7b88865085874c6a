#!/bin/bash
cd "$(dirname "$0")/.."
SNNHIP_LIB_PATH=$PWD/build/abl/libsnnhip_stemtrace.so python tools/bench_layers.py --fp16 --only=adhoc --shape 16,720,1280,3,32,9,1 --reps 2 2>&1 | grep "stemtrace" | head -12
