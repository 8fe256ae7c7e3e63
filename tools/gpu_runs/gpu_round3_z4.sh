#!/bin/bash
# round 3: 64-chain-block kernel with the phase-3 operands read one group ahead
mkdir -p gpurun_out/r03z
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py -x -q 2>&1 | tail -3
timeout 300 python tools/lb_phase_timing.py build/variants/libzshmc_lbtiming.so 256 32768 50000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z/lb_phase_timing_v2_prefetch.txt
cp zhusuan_amd/lib/libzshmc.so /tmp/libzshmc_main.so
for v in main nopre main nopre; do
  [ $v = nopre ] && cp build/variants/libzshmc_nopre.so zhusuan_amd/lib/libzshmc.so || cp /tmp/libzshmc_main.so zhusuan_amd/lib/libzshmc.so
  echo "== $v"
  timeout 300 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=256" | cut -c1-200
  timeout 300 python tools/lntm_docmajor_bench.py 1024 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/r03z/prefetch3_ab.txt
cp /tmp/libzshmc_main.so zhusuan_amd/lib/libzshmc.so
