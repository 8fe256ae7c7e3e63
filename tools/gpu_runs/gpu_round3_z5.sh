#!/bin/bash
# round 3: which phases of the 64-chain-block kernel read their operands a group ahead
mkdir -p gpurun_out/r03z
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py -x -q 2>&1 | tail -3
cp zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_main.so
for v in main nopre both bonly main nopre both bonly; do
  cp build/variants/libzshmc_$v.so zhusuan_amd/lib/libzshmc.so
  echo "== $v"
  timeout 300 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=256" | cut -c1-120
  timeout 300 python tools/lntm_docmajor_bench.py 1024 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
done 2>&1 | tee gpurun_out/r03z/prefetch3_ab.txt
cp build/variants/libzshmc_main.so zhusuan_amd/lib/libzshmc.so
