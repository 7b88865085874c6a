#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 32 64 128 256; do timeout 600 python bench.py --config c4 --micro $m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 micro $m', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'])"; done
for m in 8 16; do timeout 600 python bench.py --config c5 --micro $m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 micro $m', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'])"; done
