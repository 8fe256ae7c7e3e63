#!/bin/bash
# A/B builds of the bf16x3 likelihood kernel (csrc/b3_kernel.h, the Bernoulli /
# multinomial translation unit csrc/linear_bf16x3.hip):
#   tools/build_b3_variants.sh TAG "-DFLAG ..." [TAG2 "..."]...
# -> build/variants/libzshmc_TAG.so (only that file is recompiled; the other
# objects come from build/obj).  Switches the source knows: -DZS_B3_NACC=1|2
# (accumulator chains of GEMM 1), -DZS_B3_SWIZZLE=0|1 (chunk placement of the
# tile image).  Time with  LB_LIB=build/variants/libzshmc_TAG.so python tools/b3_bench.py
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
mkdir -p build/variants build/obj
python -c "import __graft_entry__ as g; g.build()" >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  d=build/variants/obj_$tag; mkdir -p $d
  $HIPCC $FLAGS $extra -c zhusuan_amd/csrc/linear_bf16x3.hip -o $d/linear_bf16x3.hip.o
  others=$(ls build/obj/*.o | grep -v "linear_bf16x3.hip.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/variants/libzshmc_$tag.so $d/*.o $others
  rm -rf $d
  echo "built build/variants/libzshmc_$tag.so ($extra)"
done
