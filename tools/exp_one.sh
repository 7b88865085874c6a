#!/bin/bash
# Experiment build of ONE translation unit into build/abl/libsnnhip_<tag>.so for same-box A/B runs (SNNHIP_LIB_PATH=...).
#   usage: tools/exp_one.sh <file.hip> <tag>:<flags> [<tag>:<flags> ...]
set -e
cd "$(dirname "$0")/.."
src=$1; shift
base=$(basename "$src")
mkdir -p build/abl
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $flags -c "shadernn_amd/csrc/$base" -o "build/abl/one_$tag.o"
  objs=$(ls build/obj/*.hip.o | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/abl/libsnnhip_$tag.so" $objs "build/abl/one_$tag.o" -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libsnnhip.so
  rm "build/abl/one_$tag.o"
done
ls build/abl/
