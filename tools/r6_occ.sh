#!/bin/bash
# irb_band_kernel residency experiment (round 6): MobileNetV2 b02 / b04 at batch 256 with the product geometry, with extra LDS that leaves ONE block per CU,
# and with pinned wave counts.   tools/gpu.sh <tag> sh:r6_occ.sh
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" SNNHIP_IRB_OCC=1 python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only $B 2>&1 | grep -v amdgpu.ids | cut -c1-220; }
for B in b02 b04 b03; do
  run A=0
  run SNNHIP_IRB_BAND_LDS_PAD=8192
  run A=0
done
B=b02
run SNNHIP_IRB_BAND_GEOM=8,28,8
run SNNHIP_IRB_BAND_GEOM=8,28,4
run SNNHIP_IRB_BAND_GEOM=4,28,4
run SNNHIP_IRB_BAND_GEOM=4,28,7
run SNNHIP_IRB_BAND_GEOM=4,28,8
run SNNHIP_IRB_BAND_GEOM=8,14,4
run SNNHIP_IRB_BAND_GEOM=8,14,7
