#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_two_rank.py -q -x 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r02u/pytest.txt
