#!/bin/bash
# round 3: register-resident softmax-family kernel, two rows per wave
mkdir -p gpurun_out/r03y
timeout 900 python -m pytest tests/test_gpu_distributions.py tests/test_gpu_distribution_shapes.py tests/test_gpu_hmc_reference.py tests/test_gpu_linear_bernoulli.py tests/test_gpu_native_plan_limits.py -x -q 2>&1 | tail -3
timeout 600 python tools/generic_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03y/generic_bench.txt | head -14
