#!/bin/bash
mkdir -p gpurun_out/r02j
O=gpurun_out/r02j
timeout 600 python tools/generic_bench.py > $O/generic_bench.txt 2>&1; grep -v amdgpu $O/generic_bench.txt
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_gpu_mvn.py tests/test_gpu_gather_dot.py tests/test_gpu_sgmcmc.py -m gpu -q 2>&1 | tail -4
