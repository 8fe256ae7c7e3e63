#!/bin/bash
# SQ wait/active breakdown of the fused kernel in one library build:
#   tools/pmc_kbench.sh <tag> <lib.so>   -> gpurun_out/prof/<tag>_summary.txt
TAG=$1; LIB=$2
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO python $REPO/tools/kbench.py $REPO/$LIB"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/${TAG}_pmc_sq -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/${TAG}_pmc_sq2 -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc_sq3 -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_sq3.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT $TAG
