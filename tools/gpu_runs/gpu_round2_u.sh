#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_examples.py -q -k plain_c 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r02u/pytest.txt
gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/diag_gaussian_hmc.c -Lzhusuan_amd/lib -lzshmc -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/zhusuan_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/hmc_c && /tmp/hmc_c
