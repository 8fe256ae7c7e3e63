#!/bin/bash
# round 3: wide kernel with two X-slice sets at D = 512
mkdir -p gpurun_out/r03x
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_native_plan_limits.py -x -q 2>&1 | tail -3
timeout 300 python tools/lb_wide_bench.py 8192 65536 2>&1 | grep "D=" | tee gpurun_out/r03x/lb_wide_bench.txt
timeout 300 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=" | tee -a gpurun_out/r03x/lb_wide_bench.txt
