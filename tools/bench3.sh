#!/bin/bash
# three back-to-back headline runs: images/s and the per-kernel average launch durations (us)
for i in 1 2 3; do python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
k=d.get('kernels') or d.get('config',{}).get('kernels')
print(round(d['value']), [round(x.get('avg_us',0),1) if isinstance(x,dict) else x for x in (k or [])])"; done
