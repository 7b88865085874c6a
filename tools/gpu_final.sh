#!/bin/bash
# End-of-round run on the GPU box: the whole -m gpu suite, rocprofv3 kernel trace + PMC passes of the c2 / c3 / c4 / c5 bench commands (profiles/),
# and one bench line per BASELINE config.   usage (via gpurun): tools/gpu_final.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-final}
O=gpurun_out/$TAG
mkdir -p $O
export SNN_GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 900 tools/profile_gpu.sh ${TAG}_c2 > $O/profile_c2.log 2>&1
grep "derived" -A6 $O/profile_c2.log | cut -c1-260
PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --config c3 --no-cpu-baseline --steps 10 --warmup 2 --preheat-ms 20" timeout 900 tools/profile_gpu.sh ${TAG}_c3 > $O/profile_c3.log 2>&1
PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --config c4 --no-cpu-baseline --steps 4 --warmup 1 --preheat-ms 20" timeout 900 tools/profile_gpu.sh ${TAG}_c4 > $O/profile_c4.log 2>&1
PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --config c5 --no-cpu-baseline --steps 2 --warmup 1 --preheat-ms 20" timeout 1200 tools/profile_gpu.sh ${TAG}_c5 > $O/profile_c5.log 2>&1
grep "derived" -A40 $O/profile_c5.log | cut -c1-220 | head -24
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for c in c1 c3 c4 c5; do timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<PY
import json
for c in ("c2","c1","c3","c4","c5"):
    try:
        d=json.load(open("$O/bench_%s.json"%c)); r=d.get("roofline",{}); cb=d.get("cpu_baseline",{})
        print(c, round(d["value"],1), "img/s", round(d["ms_per_step"],4), "ms/step; unfused-accounting roofline frac", round(d["frac_of_whole_step_roofline"],3), "fused", round(d.get("frac_of_sum_of_launch_rooflines",0),3),
              "| dominant", r.get("kernel","")[:50], r.get("bound"), round(r.get("frac",0),3), "traffic", r.get("traffic"), "| cpu", round(cb.get("value",0),2), cb.get("cores"))
    except Exception as e:
        print(c, "failed", e)
PY
