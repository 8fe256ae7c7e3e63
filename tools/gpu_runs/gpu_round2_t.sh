#!/bin/bash
mkdir -p gpurun_out/r02t
{
timeout 300 python tools/first_run_probe.py adapt=1
timeout 300 python tools/first_run_probe.py adapt=0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02t/first_run_probe.txt
