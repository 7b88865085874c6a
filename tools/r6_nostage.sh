#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for t in "" nostage "" nostage; do
  if [ -n "$t" ]; then export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so"; else unset SNNHIP_LIB_PATH; fi
  printf "[%s] " "${t:-product}"; python tools/bench_irb.py --batch 256 --fused-only --reps 30 --only b01 2>/dev/null | sed 's/.*fused\[//' | cut -c1-110
done
