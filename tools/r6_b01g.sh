#!/bin/bash
# round 6, split-precision kernels: MobileNetV2's b01 on irb_wave_kernel with 2x8 / 4x8 / 8x8 tiles per wave (SNNHIP_IRB_WAVE_G) and on irb_band_kernel at its best geometries
cd "$GRAFT_REPO_ROOT"
for g in 1 2 4 1 2; do printf "[wave G=%s] " $g; SNNHIP_IRB_WAVE_G=$g python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only b01 2>/dev/null | sed 's/.*fused\[//' | cut -c1-120; done
for g in 4,56,8 4,56,7 6,56,8 8,56,8 4,56,8; do printf "[band %s] " $g; SNNHIP_IRB_BAND=1 SNNHIP_IRB_BAND_GEOM=$g python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only b01 2>/dev/null | sed 's/.*fused\[//' | cut -c1-120; done
