#!/bin/bash
# Builds ablation variants of libsnnhip.so (conv2d_mfma.hip compiled with -DSNNHIP_ABL=n, everything else from the normal objects) into
# build/abl/, for tools/bench_layers.py / bench_models.py runs with SNNHIP_LIB_PATH=... (results are WRONG by construction: timing only).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
OBJ=build/obj
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DSNNHIP_ABL=$n -c shadernn_amd/csrc/conv2d_mfma.hip -o build/abl/conv2d_mfma_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls $OBJ/*.o | grep -v conv2d_mfma.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsnnhip_abl$n.so $objs build/abl/conv2d_mfma_abl$n.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
done
ls -la build/abl/*.so
