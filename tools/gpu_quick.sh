#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -12 | cut -c1-250
timeout 600 python bench.py --config c3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'], d['config']['path'][-60:])"
timeout 600 python bench.py --config c4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4', round(d['value']), d['ms_per_step'], d['frac_of_whole_step_roofline'], d['config']['path'][-60:])"
