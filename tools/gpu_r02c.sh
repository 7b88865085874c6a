#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wino_gpu.py -m gpu -q --timeout 600 -x > $O/pytest_wino.txt 2>&1
tail -25 $O/pytest_wino.txt
timeout 300 python tools/bench_layers.py --only "3x3" > $O/layers_wino.txt 2>/dev/null
SNNHIP_CONV_WINO=0 timeout 300 python tools/bench_layers.py --only "3x3" > $O/layers_direct.txt 2>/dev/null
echo "--- wino"; cut -c1-230 $O/layers_wino.txt; echo "--- direct"; cut -c1-160 $O/layers_direct.txt
timeout 600 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print('c3', d['value'], d['ms_per_step'], d['frac_of_whole_step_roofline'])
for k in d['kernels'][:12]: print('   %8.1f us x%d %6.1f TF | %s'%(k['avg_us'],k['launches'],k['flops']/k['avg_us']/1e6,k['kernel'][:150]))"
