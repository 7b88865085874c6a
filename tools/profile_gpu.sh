#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# PMC counters are collected in their own passes with kernel-trace only (never with sys/hip/hsa tracing).
# usage: tools/profile_gpu.sh <tag> [bench args...]      (PROF_CMD="python /root/repo/tools/bench_layers.py ..." profiles another command)
set -u
TAG=${1:-prof}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=${PROF_CMD:-"python $REPO/bench.py --no-cpu-baseline $*"}   # same steps/warmup as the default bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_fetch --pmc FETCH_SIZE -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_write --pmc WRITE_SIZE -- $BENCH > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
