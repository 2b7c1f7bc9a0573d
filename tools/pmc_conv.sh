#!/bin/bash
# PMC counters for conv cases (separate passes; no --stats/--sys-trace with --pmc).  Usage: tools/pmc_conv.sh tag cases ops [variants] [plans]
TAG=$1; CASES=$2; OPS=$3; VARS=${4:-}; PLANS=${5:-0}
REPO=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$TAG/set$i -o p -- python $REPO/tools/conv_bench.py --cases $CASES --ops $OPS --iters 2 --plans $PLANS ${VARS:+--variants $VARS} > $REPO/gpurun_out/pmc_$TAG/set$i.log 2>&1
done
cd $REPO
python tools/pmc_summary.py gpurun_out/pmc_$TAG
