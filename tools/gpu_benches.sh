#!/bin/bash
# one bench line per BASELINE config into gpurun_out/<tag>/ (after profiles/pmc_latest.json has been refreshed: the lines then carry `traffic`)
cd $GRAFT_REPO_ROOT
TAG=${1:-bench}
O=gpurun_out/$TAG
mkdir -p $O
export SNN_GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for c in c1 c3 c4 c5; do timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<PY
import json
for c in ("c2","c1","c3","c4","c5"):
    try:
        d=json.load(open("$O/bench_%s.json"%c)); r=d.get("roofline",{})
        print(c, round(d["value"],1), "img/s", round(d["ms_per_step"],4), "ms/step | dominant", r.get("kernel","")[:60], r.get("bound"), "frac", round(r.get("frac",0),3), "traffic", r.get("traffic"), "achieved", r.get("achieved"))
    except Exception as e:
        print(c, "failed", e)
PY
