#!/bin/bash
# round 3: first run of the feature-split (wide) likelihood kernel
mkdir -p gpurun_out/r03s
timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py -x -q -k "float64_reference or row_range" 2>&1 | tail -15
timeout 300 python tools/lb_wide_bench.py 8192 65536 2>&1 | tail -5
