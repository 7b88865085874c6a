#!/bin/bash
# Round 6: conv2d_s2march on Candy's two down-sampling layers (16 images, fp16), product build and -- lib:s2t -- the phase-trace build.
cd "$(dirname "$0")/.."
python tools/bench_layers.py --fp16 --shape 16,720,1280,32,64,3,2 --shape 16,360,640,64,128,3,2 --only adhoc --reps ${1:-50} 2>&1 | grep -v amdgpu.ids | cut -c1-260 | head -${2:-60}
