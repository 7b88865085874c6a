#!/bin/bash
# irb_band_kernel geometry re-sweep on the round-6 kernel (MobileNetV2 b02 / b03 / b04 / b06 at batch 256): tools/gpu.sh <tag> sh:r6_geom.sh
cd "$GRAFT_REPO_ROOT"
one() { printf "%s [%s] " $1 $2; SNNHIP_IRB_BAND=1 SNNHIP_IRB_BAND_GEOM=$2 python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only $1 2>/dev/null | sed 's/.*fused\[//' | cut -c1-120; }
for g in 8,28,7 8,28,8 7,28,8 7,28,7 6,28,8 6,28,7 9,28,8 5,28,8 14,14,8 14,14,7 16,14,8 12,14,8 10,14,8 8,19,8 8,56,8 4,56,8 4,56,7 3,56,8 2,56,8 2,56,4 8,28,5 8,28,6; do one b02 $g; done
for g in 2,28,4 2,28,7 2,28,8 3,28,4 4,28,4 4,28,7 4,28,8 7,28,8 4,14,4 7,14,4 7,14,7 7,14,8 14,14,8 2,14,4 1,28,4 3,28,8 5,28,8; do one b03 $g; done
for g in 7,28,8 7,28,7 4,28,7 4,28,8 4,28,4 14,14,8 14,14,7 7,14,4 14,28,8 9,28,8 8,28,8 6,28,8 5,28,8 10,14,8; do one b04 $g; done
for g in 4,14,4 7,14,4 7,14,7 7,14,8 14,14,8 14,14,7 2,14,4 3,14,4 5,14,4; do one b06 $g; done
