#!/bin/bash
# rocprofv3 kernel trace + PMC passes of one bench_layers row:  tools/gpu_prof_layer.sh <tag> "<--only pattern>" [env...]
cd $GRAFT_REPO_ROOT
TAG=$1; PAT=$2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
export PROF_CMD="python $GRAFT_REPO_ROOT/tools/bench_layers.py --only $PAT --reps ${REPS:-20}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $PROF_CMD > $O/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/pmc_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -- $PROF_CMD > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- $PROF_CMD > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/pmc_fetch --pmc FETCH_SIZE -- $PROF_CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/pmc_write --pmc WRITE_SIZE -- $PROF_CMD > $O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py $O > $O/summary.md 2>&1
grep -v "elementwise\|copyBuffer\|^$" $O/summary.md | cut -c1-250
