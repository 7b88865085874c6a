#!/bin/bash
# same-box A/B of one bench config under environment settings: tools/ab_cfg.sh c3 "A=0" "SNN_NO_BRANCH_OVERLAP=1" ...
cd "$GRAFT_REPO_ROOT"
c=$1; shift
D=$(mktemp -d)
for r in 1 2; do
for spec in "$@"; do
  env $spec python bench.py --config $c --also none --no-cpu-baseline --layer-table 0 --detail-out $D/d.json >/dev/null 2>&1
  python -c "
import json
d=json.load(open('$D/d.json'))
print('[$spec]', '%.4f ms/step %.1f img/s |' % (d['ms_per_step'], d['value']), ' '.join('%s %.0f' % (k['function'][:18], k['us_per_step']) for k in d['kernels'][:7]), '| parity', d['parity']['ok'], d['parity']['max_abs_err'])"
done
done
rm -rf $D
