#!/bin/bash
# conv2d_stem_kernel (Candy's 9x9 3 -> 32 stem, 16 images at 720p): product (branch-free loads, counted buffer stores) vs -DSNNHIP_STEM_BRANCHY_IO (rounds 2-5), ABAB
cd "$(dirname "$0")/.."
for lib in "" stemold "" stemold; do
  echo "== ${lib:-product}"
  SNNHIP_LIB_PATH=${lib:+$PWD/build/abl/libsnnhip_$lib.so} python tools/bench_layers.py --fp16 --only=adhoc --shape 16,720,1280,3,32,9,1 --reps 30 2>&1 | grep adhoc | cut -c1-120
done
