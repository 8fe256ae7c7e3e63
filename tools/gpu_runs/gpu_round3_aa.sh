#!/bin/bash
# round 3: multinomial mode of the feature-split kernel (K up to 1 024 topics)
mkdir -p gpurun_out/r03aa
timeout 900 python -m pytest tests/test_gpu_mixture_multinomial.py tests/test_gpu_native_plan_limits.py tests/test_gpu_packed_rows.py tests/test_gpu_linear_bernoulli.py -x -q 2>&1 | tail -25
