#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 600 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print('c3', d['value'], d['ms_per_step'], d['frac_of_whole_step_roofline'], d['roofline']['kernel'][:60], d['roofline']['frac'])
for k in d['kernels'][:12]: print('   %8.1f us x%d %6.1f TF | %s'%(k['avg_us'],k['launches'],k['flops']/k['avg_us']/1e6,k['kernel'][:150]))"
timeout 300 python tools/bench_models.py --model unet 2>/dev/null | head -12
timeout 300 python tools/bench_models.py --model yolov3-tiny 2>/dev/null | head -6
