#!/bin/bash
# same-box A/B of two builds of libsnnhip.so on the c2 bench line: tools/ab_c2.sh <tagA> <tagB> [rounds]   ('' = the product library)
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 ${3:-3}); do
  for t in "$1" "$2"; do
    if [ -n "$t" ] && [ "$t" != "product" ]; then export SNNHIP_LIB_PATH=$GRAFT_REPO_ROOT/build/abl/libsnnhip_$t.so; else unset SNNHIP_LIB_PATH; fi
    python bench.py --config c2 --also none --no-cpu-baseline --layer-table 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('${t:-product}', '%.4f ms/step %.1f img/s |' % (d['ms_per_step'], d['value']), ' '.join('%s %.1f us' % (k['function'][:24], k['us_per_step']) for k in d['kernels']), '| parity', d['parity']['ok'])"
  done
done
