#!/bin/bash
# Round 6: c5 (Candy 720p fp16, 64 images) as 4 x 16 and as 2 x 32 images per step, ABAB on one box.
cd "$(dirname "$0")/.."
D=$(mktemp -d)
for r in 1 2; do for m in ${1:-16 32}; do
  python bench.py --config c5 --also none --no-cpu-baseline --layer-table 0 --micro $m --detail-out $D/d.json >/dev/null 2>&1
  python -c "
import json
d=json.load(open('$D/d.json'))
print('[micro $m]', '%.3f ms/step %.1f img/s |' % (d['ms_per_step'], d['value']), ' '.join('%s %.0f' % (k['function'][:16], k['us_per_step']) for k in d['kernels'][:6]), '| launches', d['config'].get('launches_per_step'), 'parity', d['parity']['ok'])"
done; done
rm -rf $D
