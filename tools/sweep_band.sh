#!/bin/bash
# GPU-side sweep of irb_band_kernel geometries (SNNHIP_IRB_BAND_GEOM=R,SW,NW) per MobileNetV2 block: tools/sweep_band.sh b04 "7,28,8" "4,28,7" ...
cd "$GRAFT_REPO_ROOT"
b=$1; shift
for g in "$@"; do
  SNNHIP_IRB_BAND=1 SNNHIP_IRB_BAND_GEOM=$g python tools/bench_irb.py --batch 256 --fused-only --reps 20 --only $b 2>/dev/null | sed "s/^/[$g] /" | cut -c1-200
done
