#!/bin/bash
# round-2 first GPU call: the whole -m gpu suite (new full-size config tests included) + one bench line per config
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x --deselect tests/test_configs_gpu.py > $O/pytest_main.txt 2>&1
tail -5 $O/pytest_main.txt
timeout 1200 python -m pytest tests/test_configs_gpu.py -m gpu -q --timeout 900 --durations=10 > $O/pytest_configs.txt 2>&1
tail -40 $O/pytest_configs.txt
for c in c2 c1 c3 c4 c5; do
  timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  echo "bench $c rc=$?"; head -c 1500 $O/bench_$c.json; echo; tail -3 $O/bench_$c.err
done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err
head -c 400 $O/bench_c2_driver.json
