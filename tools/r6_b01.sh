#!/bin/bash
# MobileNetV2 b01 (16 -> 96 -> 24, 112x112 -> 56x56) on irb_band_kernel again, round-6 kernel: geometries against irb_wave_kernel.  tools/gpu.sh <tag> sh:r6_b01.sh
cd "$GRAFT_REPO_ROOT"
one() { printf "[%s] " $1; SNNHIP_IRB_BAND=1 SNNHIP_IRB_BAND_GEOM=$1 python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only b01 2>/dev/null | sed 's/.*fused\[//' | cut -c1-120; }
printf "[product] "; python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only b01 2>/dev/null | sed 's/.*fused\[//' | cut -c1-120
for g in 2,28,4 2,28,8 4,28,8 4,28,4 2,56,8 2,56,4 4,56,8 3,28,8 4,14,4 4,14,8 8,14,8 2,14,4 1,56,4 4,28,7 2,28,7; do one $g; done
