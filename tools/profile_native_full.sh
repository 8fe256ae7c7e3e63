#!/bin/bash
# rocprofv3 evidence for the MFMA likelihood kernel at the FULL shapes of
# BASELINE configs[2] and configs[4]: kernel trace + separate PMC passes
# (FETCH_SIZE, WRITE_SIZE; MFMA busy).  Output: gpurun_out/prof/<tag>_nativefull_*
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO python $REPO/tools/native_kernel_pmc.py ${2:-8192}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_nativefull_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_nativefull_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_nativefull_fetch -o pmc --output-format csv -- $CMD > $OUT/${TAG}_nativefull_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_nativefull_write -o pmc --output-format csv -- $CMD > $OUT/${TAG}_nativefull_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_nativefull_mfma -o pmc --output-format csv -- $CMD > $OUT/${TAG}_nativefull_mfma.log 2>&1
cd $REPO
python tools/summarize_native_full.py $OUT $TAG
