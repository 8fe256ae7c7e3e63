#!/bin/bash
# round 3: phase-3 read-ahead of the 64-chain-block kernel held in place by a
# scheduling fence (hipcc sank the reads back to their uses)
mkdir -p gpurun_out/r03dd
cp zhusuan_amd/lib/libzshmc.so /tmp/main.so
for v in main fence fenceb main fence fenceb; do
  [ $v = main ] && cp /tmp/main.so zhusuan_amd/lib/libzshmc.so || cp build/variants/libzshmc_$v.so zhusuan_amd/lib/libzshmc.so
  echo "== $v"
  timeout 300 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=256" | cut -c1-120
  timeout 300 python tools/lntm_docmajor_bench.py 1024 2>&1 | tail -1 | cut -c1-150
done 2>&1 | tee gpurun_out/r03dd/prefetch_fence_ab.txt
cp /tmp/main.so zhusuan_amd/lib/libzshmc.so
