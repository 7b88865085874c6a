set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01fin6
mkdir -p $O
timeout 900 tools/profile_gpu.sh r01fin6 > $O/profile.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_layers.py > $O/layers.txt 2>/dev/null
timeout 300 python tools/bench_layers.py --fp16 > $O/layers_fp16.txt 2>/dev/null
: > $O/models.txt
for m in resnet18 mobilenetv2 candy unet yolov3-tiny; do timeout 300 python tools/bench_models.py --model $m 2>/dev/null | grep -v amdgpu >> $O/models.txt; done
timeout 300 python tools/bench_models.py --model candy --batch 8 2>/dev/null | grep -v amdgpu >> $O/models.txt
: > $O/models_fp16.txt
for m in resnet18 mobilenetv2 candy unet yolov3-tiny; do timeout 300 python tools/bench_models.py --model $m --fp16 2>/dev/null | grep -v amdgpu >> $O/models_fp16.txt; done
timeout 300 python tools/bench_models.py --model candy --batch 8 --fp16 2>/dev/null | grep -v amdgpu >> $O/models_fp16.txt
: > $O/models_tuned.txt
for f in "" "--fp16"; do for m in resnet18 mobilenetv2 candy unet yolov3-tiny; do timeout 300 python tools/bench_models.py --model $m $f --tune 2>/dev/null | grep -v amdgpu | sed -n 1,3p >> $O/models_tuned.txt; done; done
./build/ubench_mfma_peak > $O/ubench_mfma_peak.txt 2>&1
./build/ubench_valu_peak > $O/ubench_valu_peak.txt 2>&1
tail -c 600 $O/bench.json
