#!/bin/bash
# irb_band_kernel wave-priority experiment (round 6): tools/exp_one.sh irb_fused.hip prio1:-DSNNHIP_IRBB_PRIO=1 prio0:-DSNNHIP_IRBB_PRIO=0; tools/gpu.sh <tag> sh:r6_prio.sh
cd "$GRAFT_REPO_ROOT"
for t in "" prio1 prio0 "" prio1 prio0; do
  if [ -n "$t" ]; then export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so"; else unset SNNHIP_LIB_PATH; fi
  for b in b02 b03 b04 b06; do printf "[%s] " "${t:-product}"; python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only $b 2>/dev/null | sed 's/.*fused\[//' | cut -c1-110; done
done
