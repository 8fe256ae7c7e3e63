#!/bin/bash
# A/B builds of libzshmc.so: tools/build_variants.sh TAG "-DFLAG ..." [TAG2 "..."]...
# -> build/variants/libzshmc_TAG.so (only the two fused-kernel files are
# recompiled with the extra flags; everything else comes from build/obj).
# tools/kbench.py then times them side by side on the GPU.
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
mkdir -p build/variants build/obj
python -c "import __graft_entry__ as g; g.build()" >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  d=build/variants/obj_$tag; mkdir -p $d
  for f in hmc_fused_ring hmc_fused_normal; do
    $HIPCC $FLAGS $extra -c zhusuan_amd/csrc/$f.hip -o $d/$f.hip.o &
  done
  wait
  others=$(ls build/obj/*.o | grep -v "hmc_fused_ring\|hmc_fused_normal")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/variants/libzshmc_$tag.so $d/*.o $others
  rm -rf $d
  echo "built build/variants/libzshmc_$tag.so ($extra)"
done
