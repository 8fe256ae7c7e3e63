#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box via gpurun):
#   1. --kernel-trace --stats  (per-kernel durations)
#   2. separate --pmc passes   (HBM bytes; VALU / wait breakdown)
# Output: gpurun_out/prof/<tag>_*   (copy the summaries into profiles/).
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-ess --no-extra-configs"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/${TAG}_pmc_sq -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_sq2 -o pmc --output-format csv -- $CMD > $OUT/${TAG}_pmc_sq2.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT $TAG
