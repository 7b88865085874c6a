#!/bin/bash
# same-box A/B of two builds of libsnnhip.so over the benchmark graphs: tools/ab_models.sh <old.so> [<new.so>]
OLD=$1; NEW=${2:-shadernn_amd/lib/libsnnhip.so}
for m in resnet18 mobilenetv2 yolov3-tiny unet candy; do for f in "" "--fp16"; do
  a=$(SNNHIP_LIB_PATH=$OLD timeout 200 python tools/bench_models.py --model $m $f 2>/dev/null | grep -o "batch.*ms/batch" | head -1)
  b=$(SNNHIP_LIB_PATH=$NEW timeout 200 python tools/bench_models.py --model $m $f 2>/dev/null | grep -o "batch.*ms/batch" | head -1)
  a2=$(SNNHIP_LIB_PATH=$OLD timeout 200 python tools/bench_models.py --model $m $f 2>/dev/null | grep -o "batch.*ms/batch" | head -1)
  b2=$(SNNHIP_LIB_PATH=$NEW timeout 200 python tools/bench_models.py --model $m $f 2>/dev/null | grep -o "batch.*ms/batch" | head -1)
  echo "$m $f | old: $a / $a2 | new: $b / $b2"
done; done
