#!/bin/bash
# round 3: feature-split kernel with phase 3 hand-pipelined
mkdir -p gpurun_out/r03z
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_native_plan_limits.py -x -q 2>&1 | tail -3
for D in 1024 512; do
  timeout 300 python tools/lbw_phase_timing.py build/variants/libzshmc_lbwtiming.so $D 8192 16384
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z/lbw_phase_timing_pipelined.txt
timeout 300 python tools/lb_wide_bench.py 8192 65536 2>&1 | grep "D=" | tee gpurun_out/r03z/lb_wide_bench_pipelined.txt
