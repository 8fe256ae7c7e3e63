#!/bin/bash
# A/B builds of the two-GEMM likelihood kernels:
#   tools/build_lb_variants.sh TAG "-DFLAG ..." [TAG2 "..."]...
# -> build/variants/libzshmc_TAG.so (only csrc/linear_bernoulli*.hip are
# recompiled with the extra flags; the other
# objects come from build/obj).  Compile-time switches the sources know:
# -DZS_LB_TIMING (per-phase shader clocks, tools/lb_phase_timing.py),
# -DZS_LB_LDS_PAD=bytes (extra LDS: fewer workgroups per CU at D <= 128).
# Time them with
#   LB_LIB=build/variants/libzshmc_TAG.so python tools/lb_modes_bench.py
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
mkdir -p build/variants build/obj
python -c "import __graft_entry__ as g; g.build()" >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  d=build/variants/obj_$tag; mkdir -p $d
  for f in linear_bernoulli linear_bernoulli_mid linear_bernoulli_wide; do
    $HIPCC $FLAGS $extra -c zhusuan_amd/csrc/$f.hip -o $d/$f.hip.o &
  done
  wait
  others=$(ls build/obj/*.o | grep -v "linear_bernoulli")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/variants/libzshmc_$tag.so $d/*.o $others
  rm -rf $d
  echo "built build/variants/libzshmc_$tag.so ($extra)"
done
