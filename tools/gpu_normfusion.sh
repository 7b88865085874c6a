#!/bin/bash
# chain rule F (tile statistics from conv2d_wide_f16 instead of the InstanceNorm's statistics sweep): Candy 720p fp16 with / without, per batch size
cd $GRAFT_REPO_ROOT
for b in ${@:-1 2 4 8}; do
  for f in 0 x; do
    if [ $f = 0 ]; then export SNNHIP_NORM_FUSION=0; unset SNNHIP_NORM_FUSION_MIN_MB; else unset SNNHIP_NORM_FUSION; export SNNHIP_NORM_FUSION_MIN_MB=0; fi
    echo "batch $b rule F $( [ $f = 0 ] && echo off || echo on ): $(timeout 300 python tools/bench_models.py --model candy --fp16 --batch $b 2>/dev/null | sed -n 2p | cut -c1-90)"
  done
done
