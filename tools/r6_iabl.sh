#!/bin/bash
# irb_image_kernel ablation (round 6): the product against builds without the E epilogue (abl1), without D's tap reads + FMAs (abl2), without both (abl3: the
# MFMA skeleton + staging + reduce).  Built HERE by tools/exp_one.sh irb_fused.hip abl1:-DSNNHIP_IRBI_ABL=1 abl2:-DSNNHIP_IRBI_ABL=2 abl3:-DSNNHIP_IRBI_ABL=3
cd "$GRAFT_REPO_ROOT"
for t in "" abl1 abl2 abl3 ""; do
  if [ -n "$t" ]; then export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so"; else unset SNNHIP_LIB_PATH; fi
  for b in b07 b10 b11 b13 b14; do
    printf "[%s] " "${t:-product}"; python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only $b 2>/dev/null | cut -c1-140
  done
done
