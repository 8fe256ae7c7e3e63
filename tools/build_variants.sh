#!/bin/bash
# Build A/B variants of libzshmc.so into build/variants/ : one per line of
# "name  extra-hipcc-flags".  Usage: tools/build_variants.sh < variants.txt
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
while read -r name flags; do
  [ -z "$name" ] && continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
    -o build/variants/lib_$name.so zhusuan_amd/csrc/*.hip &
done
wait
ls -la build/variants/
