#!/bin/bash
# round 5: the persistent wide kernel on a GPU box -- parity first (short timeouts: a hung kernel must not eat the lease), then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r5_wide}; mkdir -p "$O"
timeout 420 python -m pytest tests/test_conv_wide_gpu.py -m gpu -q -x --timeout 120 -k "wide" > "$O/t_wide.txt" 2>&1; rc=$?; echo "wide tests rc=$rc"; tail -25 "$O/t_wide.txt" | cut -c1-220
if [ $rc -ne 0 ]; then exit 0; fi
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_fp16_gpu.py tests/test_guard_gpu.py -m gpu -q --timeout 600 -k "c5 or candy or guard or style" > "$O/t_c5.txt" 2>&1; echo "c5 tests rc=$?"; tail -8 "$O/t_c5.txt" | cut -c1-220
tools/ab_wide.sh SNNHIP_WIDE_PERSIST=0 SNNHIP_WIDE_PERSIST=1 > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
for spec in SNNHIP_WIDE_PERSIST=0 SNNHIP_WIDE_PERSIST=1; do
  env $spec timeout 600 python bench.py --config c5 --also none --no-cpu-baseline --layer-table 0 > "$O/bench_c5_$spec.json" 2> "$O/bench_c5_$spec.err" || tail -3 "$O/bench_c5_$spec.err"
  python tools/bench_digest.py "$O/bench_c5_$spec.json" | head -8
done
