#!/bin/bash
# rocprofv3 evidence for the two-GEMM likelihood kernels across widths: kernel
# trace of tools/lb_modes_bench.py + a separate MFMA-busy PMC pass.
#   bash tools/profile_lb_modes.sh TAG "64,256,512,832,1024"
# Output: gpurun_out/prof/<tag>_lbmodes_*  (copy the summary into profiles/).
TAG=${1:-r04}
WIDTHS=${2:-256,512,832,1024}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO LB_WIDTHS=$WIDTHS python $REPO/tools/lb_modes_bench.py"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_lbmodes_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_lbmodes_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${TAG}_lbmodes_pmc -o pmc --output-format csv -- $CMD > $OUT/${TAG}_lbmodes_pmc.log 2>&1
cd $REPO
python tools/summarize_lb_modes_prof.py $OUT $TAG
