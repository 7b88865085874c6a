#!/bin/bash
# per-kernel table of c5 (Candy 720p fp16) at a given micro-batch: tools/gpu_c5_micro.sh <micro> [rows]
cd $GRAFT_REPO_ROOT
M=${1:-1}
timeout 600 python bench.py --config c5 --micro $M --no-cpu-baseline --all-kernels 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 micro $M', round(d['value']), d['ms_per_step'])
tot=sum(k['avg_us']*k['launches'] for k in d['kernels'])
print('sum of kernels %.1f us over %d launches'%(tot, sum(k['launches'] for k in d['kernels'])))
for k in d['kernels'][:${2:-40}]: print('   %8.1f us x%d | %s'%(k['avg_us'],k['launches'],k['kernel'][:150]))"
