#!/bin/bash
# same-box runs of the 128 -> 128 body layer of Candy (tools/bench_layers.py) under switch settings: tools/ab_wide.sh "ENV=1 ENV2=x" "..." ...
cd "$GRAFT_REPO_ROOT"
L="--fp16 --only adhoc --shape 16,183,323,128,128,3,1 --shape 16,408,688,128,64,3,1"
for r in 1 2; do
for spec in "$@"; do
  echo "== [$spec]"
  env $spec python tools/bench_layers.py $L 2>/dev/null | grep adhoc | cut -c1-100
done
done
