#!/bin/bash
# Builds ablation variants of libsnnhip.so (the conv2d_mfma kernel translation units compiled with -DSNNHIP_ABL=n, everything else from the
# normal objects) into build/abl/, for tools/bench_layers.py / bench_models.py runs with SNNHIP_LIB_PATH=... (results are WRONG by
# construction: timing only).   usage: tools/ablate_conv.sh 1 2 3 ...   (bits: 1 no weight refills, 2 no LDS operand reads, 4 no activation
# loads, 8 no output stores)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
OBJ=build/obj
for n in "$@"; do
  for v in 128_f32 128_f16 64_f32 64_f16 32_f32 32_f16; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DSNNHIP_ABL=$n -c shadernn_amd/csrc/conv2d_mfma_bn$v.hip -o build/abl/conv2d_mfma_bn${v}_abl$n.o &
  done
done
wait
for n in "$@"; do
  objs=$(ls $OBJ/*.o | grep -v "conv2d_mfma_bn")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsnnhip_abl$n.so $objs build/abl/conv2d_mfma_bn*_abl$n.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
done
ls -la build/abl/*.so
