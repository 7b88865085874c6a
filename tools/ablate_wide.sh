#!/bin/bash
# Ablation builds of conv2d_wide_f16.hip (-DSNNHIP_WIDE_ABL=n; results are WRONG by construction, timing only) into build/abl/ for
# SNNHIP_LIB_PATH=build/abl/libsnnhip_wide<n>.so runs.   usage: tools/ablate_wide.sh 1 2 4 8 15
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DSNNHIP_WIDE_ABL=$n -c shadernn_amd/csrc/conv2d_wide_f16.hip -o build/abl/wide_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/obj/*.hip.o | grep -v "conv2d_wide_f16")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsnnhip_wide$n.so $objs build/abl/wide_abl$n.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
  rm build/abl/wide_abl$n.o
done
ls -la build/abl/*.so
