#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_irb_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8 | cut -c1-250
python tools/bench_irb.py --batch 256 --reps 5 | cut -c1-130
for e in "" "SNNHIP_NO_IRB_FUSION=1"; do
env $e timeout 600 python bench.py --config c4 --no-cpu-baseline --through capi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 $e', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'])"
done
