#!/bin/bash
# round 3: 64-chain-block kernel at D = 256, next tile's DMA under phase 3b
# (main) against under phase 1 (dmap1 = the form of rounds 1-2)
mkdir -p gpurun_out/r03cc
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py -x -q -k "float64 or row_range or config3_full or document_major" 2>&1 | tail -2
cp zhusuan_amd/lib/libzshmc.so /tmp/main.so
for v in main dmap1 main dmap1; do
  [ $v = dmap1 ] && cp build/variants/libzshmc_dmap1.so zhusuan_amd/lib/libzshmc.so || cp /tmp/main.so zhusuan_amd/lib/libzshmc.so
  echo "== $v"
  timeout 300 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=256" | cut -c1-170
  timeout 300 python tools/lntm_docmajor_bench.py 256 256 2>&1 | tail -1 | cut -c1-150
done 2>&1 | tee gpurun_out/r03cc/dma_phase3_ab.txt
cp /tmp/main.so zhusuan_amd/lib/libzshmc.so
