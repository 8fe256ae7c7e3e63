#!/bin/bash
# rocprofv3 evidence for the bf16x3 likelihood kernels (csrc/b3_kernel.h)
# beside the exact-fp32 ones: kernel trace of tools/b3_bench.py + a separate
# MFMA-busy PMC pass (counters in their own run, kernel trace / stats only).
#   bash tools/profile_b3.sh TAG "128,256"
# Output: gpurun_out/prof/<tag>_b3_*  (copy the summary into profiles/).
TAG=${1:-r05}
WIDTHS=${2:-128,256}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO B3_WIDTHS=$WIDTHS python $REPO/tools/b3_bench.py 2"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_b3_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_b3_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${TAG}_b3_pmc -o pmc --output-format csv -- $CMD > $OUT/${TAG}_b3_pmc.log 2>&1
cd $REPO
python tools/summarize_b3_prof.py $OUT $TAG
