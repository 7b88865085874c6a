#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
for c in c3 c4 c5; do
  for t in host capi; do
    timeout 600 python bench.py --config $c --through $t --no-cpu-baseline > $O/bench_${c}_$t.json 2> $O/bench_${c}_$t.err
    echo "bench $c $t rc=$?"; python -c "
import json,sys
d=json.load(open('$O/bench_${c}_$t.json'))
print(d['value'], d['ms_per_step'], d['config']['path'][:150]); print(d.get('roofline',{}).get('kernel','')[:120], d.get('roofline',{}).get('frac'))
" ; tail -3 $O/bench_${c}_$t.err
  done
done
timeout 600 python bench.py --config c3 --through host --no-capture --no-cpu-baseline > $O/bench_c3_host_nocapture.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c3_host_nocapture.json')); print('c3 host nocapture', d['value'], d['ms_per_step'])"
