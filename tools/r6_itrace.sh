#!/bin/bash
# irb_image_kernel phase trace (s_memtime stamps of block 100, waves 0 and 3) of the product and of the ablation builds:
#   tools/exp_one.sh irb_fused.hip tr0:-DSNNHIP_IRBI_TRACE=1 tr3:"-DSNNHIP_IRBI_TRACE=1 -DSNNHIP_IRBI_ABL=3" ...;  tools/gpu.sh <tag> sh:r6_itrace.sh
cd "$GRAFT_REPO_ROOT"
for t in tr0 tr1 tr2 tr3; do
  export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so"
  for b in b11 b07 b14; do
    echo "== $t $b"; python tools/bench_irb.py --batch 256 --fused-only --reps 2 --only $b 2>/dev/null | grep irbi | sort | uniq -c | sort -rn | head -4 | cut -c1-220
  done
done
