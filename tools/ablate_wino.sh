#!/bin/bash
# Ablation variants of libsnnhip.so for conv2d_wino.hip (-DSNNHIP_WINO_ABL=n; results WRONG by construction: timing only) into build/abl/.
# usage: tools/ablate_wino.sh 1 2 4 ...   bits: 1 no barrier, 2 no LDS stores, 4 no global loads, 8 no input transform, 16 no U reads, 32 no patch reads
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
OBJ=build/obj
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DSNNHIP_WINO_ABL=$n -c shadernn_amd/csrc/conv2d_wino.hip -o build/abl/conv2d_wino_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls $OBJ/*.o | grep -v "conv2d_wino\|host_")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsnnhip_wabl$n.so $objs build/abl/conv2d_wino_abl$n.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
done
ls build/abl/*wabl*.so
