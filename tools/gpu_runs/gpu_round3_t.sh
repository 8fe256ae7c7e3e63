#!/bin/bash
# round 3: the native plan beyond one latent / D % 4 == 0 / D <= 256
mkdir -p gpurun_out/r03t
timeout 900 python -m pytest tests/test_gpu_native_plan_limits.py -x -q 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py tests/test_gpu_hmc_reference.py -x -q 2>&1 | tail -15
