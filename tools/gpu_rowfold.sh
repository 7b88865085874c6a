#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fp16_gpu.py -m gpu -q --timeout 600 -k "rowfold" 2>&1 | tail -25 | cut -c1-220
timeout 300 python tools/bench_layers.py --only "candy out" --fp16 2>/dev/null | cut -c1-200
SNNHIP_CONV=thin timeout 300 python tools/bench_layers.py --only "candy out" --fp16 2>/dev/null | cut -c1-120
timeout 600 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'])
for k in d['kernels'][:6]: print('   %8.1f us x%d %6.1f TF | %s'%(k['avg_us'],k['launches'],k['flops']/k['avg_us']/1e6,k['kernel'][:150]))"
