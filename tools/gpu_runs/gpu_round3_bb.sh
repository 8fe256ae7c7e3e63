#!/bin/bash
# round 3: D = 512 of the feature-split kernel with two slice sets, the next
# tile's DMA spread under phase 1
mkdir -p gpurun_out/r03bb
cp zhusuan_amd/lib/libzshmc.so /tmp/main.so
for v in main buf2 main buf2; do
  [ $v = buf2 ] && cp build/variants/libzshmc_buf2.so zhusuan_amd/lib/libzshmc.so || cp /tmp/main.so zhusuan_amd/lib/libzshmc.so
  echo "== $v"
  timeout 300 python tools/lb_wide_bench.py 8192 65536 2>&1 | grep "D=512" | cut -c1-160
  timeout 300 python tools/lntm_docmajor_bench.py 256 512 2>&1 | tail -1 | cut -c1-150
done 2>&1 | tee gpurun_out/r03bb/buf2_ab.txt
cp build/variants/libzshmc_buf2.so zhusuan_amd/lib/libzshmc.so
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py -x -q -k "float64 or row_range or document_major" 2>&1 | tail -2
cp /tmp/main.so zhusuan_amd/lib/libzshmc.so
