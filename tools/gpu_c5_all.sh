#!/bin/bash
# one-line-per-kernel table of a bench config (default c5), all kernels of the step
cd $GRAFT_REPO_ROOT
CFG=${1:-c5}
timeout 600 python bench.py --config $CFG --no-cpu-baseline --all-kernels 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'])
tot=sum(k['avg_us']*k['launches'] for k in d['kernels'])
print('sum of kernels %.1f us'%tot)
for k in d['kernels']: print('   %8.1f us x%d %6.1f TF %6.0f GB/s | %s'%(k['avg_us'],k['launches'],k['flops']/k['avg_us']/1e6,k['bytes']/k['avg_us']/1e3,k['kernel'][:170]))"
