#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_framework.py tests/test_gpu_distribution_shapes.py -q 2>&1 | grep -v amdgpu.ids | tail -60 | tee gpurun_out/r02u/pytest.txt
