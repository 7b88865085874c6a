#!/bin/bash
# same-box ABAB of the headline config between the product library and an experiment build: tools/ab_c2_lib.sh <tag> [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  for lib in "" "$tag"; do
    if [ -n "$lib" ]; then export SNNHIP_LIB_PATH="$GRAFT_REPO_ROOT/build/abl/libsnnhip_$lib.so"; else unset SNNHIP_LIB_PATH; fi
    python bench.py --config c2 --also none --no-cpu-baseline --layer-table 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-10s %.1f img/s  %.4f ms  A %.1f us  B %.1f us' % ('${lib:-product}', d['value'], d['ms_per_step'], d['kernels'][0]['us_per_step'], d['kernels'][1]['us_per_step']))"
  done
done
