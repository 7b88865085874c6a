#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_wino_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
SNNHIP_WINO_OPB=2 timeout 900 python -m pytest tests/test_conv_wino_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
for i in 1 2; do
echo "--- OPB=1"; timeout 300 python tools/bench_layers.py --only "3x3" 2>/dev/null | grep -i "wino" | cut -c1-75
echo "--- OPB=2"; SNNHIP_WINO_OPB=2 timeout 300 python tools/bench_layers.py --only "3x3" 2>/dev/null | grep -i "wino" | cut -c1-75
done
