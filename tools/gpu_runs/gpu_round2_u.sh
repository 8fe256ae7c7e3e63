#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 1200 python -m pytest tests/test_gpu_lntm_ais.py tests/test_gpu_mixture_multinomial.py tests/test_gpu_linear_bernoulli.py tests/test_gpu_hmc_reference.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r02u/pytest.txt
timeout 300 python tools/generic_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
