#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
export SNN_GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 900 tools/profile_gpu.sh r02_c2 > $O/profile_c2.log 2>&1
tail -30 $O/profile_c2.log | cut -c1-220
PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --config c3 --no-cpu-baseline --steps 10 --warmup 2 --preheat-ms 20" timeout 900 tools/profile_gpu.sh r02_c3 > $O/profile_c3.log 2>&1
grep "derived" -A30 $O/profile_c3.log | cut -c1-220 | head -40
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:80])"
