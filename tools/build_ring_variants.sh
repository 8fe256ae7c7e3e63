#!/bin/bash
# A/B builds of the fused transition kernel (csrc/hmc_fused_ring.hip):
#   tools/build_ring_variants.sh TAG "-DFLAG ..." [TAG2 "..."]...
# -> build/variants/libzshmc_TAG.so (only that file is recompiled; the other
# objects come from build/obj).  Switches the source knows: -DZS_CS_SKIP=1
# (COLSTATS without the accumulation: what the extra ring slot alone costs),
# -DZS_CS_SKIP=2 (the arithmetic without the LDS atomics).
# RING_SRC=<file>: another version of the source (e.g. `git show HEAD:... >
# /tmp/ring_head/hmc_fused_ring.hip`, next to copies of the headers).
# Time with  tools/ring_variant_stats.sh build/variants/libzshmc_TAG.so ...
set -e
cd "$(dirname "$0")/.."
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
mkdir -p build/variants build/obj
python -c "import __graft_entry__ as g; g.build()" >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  d=build/variants/obj_$tag; mkdir -p $d
  $HIPCC $FLAGS $extra -c ${RING_SRC:-zhusuan_amd/csrc/hmc_fused_ring.hip} -o $d/hmc_fused_ring.hip.o
  others=$(ls build/obj/*.o | grep -v "hmc_fused_ring.hip.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/variants/libzshmc_$tag.so $d/*.o $others
  rm -rf $d
  echo "built build/variants/libzshmc_$tag.so ($extra)"
done
