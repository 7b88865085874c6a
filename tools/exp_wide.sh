#!/bin/bash
# Experiment builds of conv2d_wide_f16.hip into build/abl/libsnnhip_<tag>.so for same-box A/B runs (SNNHIP_LIB_PATH=...).
#   usage: tools/exp_wide.sh a4:-DSNNHIP_WIDE_ABL=4 a15:-DSNNHIP_WIDE_ABL=15 mine:-DMY_EXPERIMENT=1 ...   (one library per tag:flags pair)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $flags -c shadernn_amd/csrc/conv2d_wide_f16.hip -o build/abl/wide_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%:*}
  objs=$(ls build/obj/*.hip.o | grep -v "conv2d_wide_f16")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libsnnhip_$tag.so $objs build/abl/wide_$tag.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
  rm build/abl/wide_$tag.o
done
ls build/abl/
