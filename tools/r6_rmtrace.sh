#!/bin/bash
cd "$(dirname "$0")/.."
SNNHIP_LIB_PATH=$PWD/build/abl/libsnnhip_rmtrace.so python tools/bench_layers.py --fp16 --only=adhoc --shape 16,728,1288,32,3,9,1 --reps 2 2>&1 | grep -i "rmepi" | head -12
