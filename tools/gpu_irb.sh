#!/bin/bash
# inverted-residual blocks: parity tests, then per-block timings (separate layers / the fused kernel at each tile size per wave)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_irb_gpu.py -m gpu -q --timeout 600 2>&1 | tail -3 | cut -c1-250
SNNHIP_IRB_FUSION=all python tools/bench_irb.py --batch ${1:-256} --reps 5 | cut -c1-130
for g in 4 2 1; do
echo "== tile per wave: G=$g"; SNNHIP_IRB_FUSION=all SNNHIP_IRB_WAVE_G=$g python tools/bench_irb.py --batch ${1:-256} --reps 5 --fused-only | cut -c1-110
done
