#!/bin/bash
# conv2d_rowfold_march_kernel (Candy's 9x9 32 -> 3 output layer, 16 images at 720p behind its reflect pad): product library vs build/abl/libsnnhip_pulls.so (the ds_bpermute epilogue), ABAB
cd "$(dirname "$0")/.."
for r in 1 2; do
  for lib in "" "$PWD/build/abl/libsnnhip_pulls.so"; do
    echo "== ${lib:-product}"
    SNNHIP_LIB_PATH=$lib python tools/bench_layers.py --fp16 --only=adhoc --shape 16,728,1288,32,3,9,1 --reps 30 2>&1 | grep adhoc | cut -c1-200
  done
done
