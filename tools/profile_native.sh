#!/bin/bash
# rocprofv3 evidence for the native dense-likelihood plans (configs 3 and 5):
#   1. --kernel-trace --stats of tools/native_plan_trace.py, cut to the marked
#      transitions: which kernels a transition launches (no at::native::*),
#      and the MFMA kernels' average durations;
#   2. a separate --pmc pass: MFMA busy cycles of both modes of the kernel.
# Output: gpurun_out/prof/<tag>_native_*  (copy the summary into profiles/).
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO python $REPO/tools/native_plan_trace.py ${2:-100000} ${3:-128}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_native_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_native_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/${TAG}_native_pmc -o pmc --output-format csv -- $CMD > $OUT/${TAG}_native_pmc.log 2>&1
cd $REPO
python tools/summarize_native.py $OUT $TAG
