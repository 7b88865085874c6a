#!/bin/bash
# whole -m gpu suite + the default bench line + one line per config into gpurun_out/$1
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-full}
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for c in c1 c3 c4 c5; do timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<PY
import json
for c in ("c2","c1","c3","c4","c5"):
    try:
        d=json.load(open("$O/bench_%s.json"%c)); r=d.get("roofline",{}); cb=d.get("cpu_baseline",{})
        print(c, round(d["value"],1), "img/s", round(d["ms_per_step"],4), "ms/step; whole-step roofline frac", round(d["frac_of_whole_step_roofline"],3), "| dominant", r.get("kernel","")[:50], r.get("bound"), round(r.get("frac",0),3), "| cpu", round(cb.get("value",0),2), cb.get("cores"))
    except Exception as e:
        print(c, "failed", e)
PY
