#!/bin/bash
# A/B builds of libzshmc.so that differ in csrc/linear_bernoulli.hip only:
#   tools/ab_lb_variants.sh TAG "-DFLAG ..." [TAG2 "..."] ...
# -> build/variants/libzshmc_TAG.so (travels to the GPU box with gpurun; on the
# box copy one over zhusuan_amd/lib/libzshmc.so, run tools/lb_wide_bench.py /
# tools/lntm_docmajor_bench.py, restore).  The switches of that file:
#   ZS_LB_PREFETCH3A(D) / ZS_LB_PREFETCH3B(D)   phase-3 operands a row group ahead
#   ZS_LB_PREFETCH_FENCE(D)                     ... held by a scheduling fence
#   ZS_LB_PREFETCH3A_INSIDE                     ... as slots inside the MFMA/VALU
#                                               pipeline of phase 3a (round 3:
#                                               126.3 against 131.6 TFLOP/s at
#                                               D = 256, tools/lb_quick_ab.py)
#   ZS_LB_DMA_PHASE3, ZS_LB_BUF(D), ZS_LB_MINW(D), ZS_LB_NO_SGB
# Example:
#   tools/ab_lb_variants.sh inside "-DZS_LB_PREFETCH3A_INSIDE=1 -DZS_LB_PREFETCH3A(D)=((D)>=128) -DZS_LB_PREFETCH_FENCE(D)=0"
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude"
mkdir -p build/variants
python -c "import __graft_entry__ as g; g.build()" >/dev/null
others=$(ls build/obj/*.o | grep -v "linear_bernoulli.hip.o")
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  # (word splitting of $extra is intended: several -D flags)
  $HIPCC $FLAGS $extra -c zhusuan_amd/csrc/linear_bernoulli.hip -o build/variants/lb_$tag.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/variants/libzshmc_$tag.so build/variants/lb_$tag.o $others
  rm -f build/variants/lb_$tag.o
  echo "built build/variants/libzshmc_$tag.so ($extra)"
done
