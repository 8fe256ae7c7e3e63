#!/bin/bash
mkdir -p gpurun_out/r02y
timeout 300 python tools/startup_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02y/startup_probe.txt
