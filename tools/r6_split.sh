#!/bin/bash
# round 6: irb_image_kernel's split-precision pointwise stages (three f16 products per fp32 product) against its fp32 MFMA form, same library, ABAB
cd "$GRAFT_REPO_ROOT"
for b in ${@:-b07 b10 b11 b13 b14}; do
  for t in 0 1 0 1; do
    printf "[%s split=%s] " "$b" "$t"; SNNHIP_IRB_SPLIT=$t python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only $b 2>/dev/null | sed 's/.*fused\[//' | cut -c1-150
  done
done
