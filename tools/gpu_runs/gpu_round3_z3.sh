#!/bin/bash
# round 3: per-phase clocks of the 64-chain-block kernel (configs[2] / [4])
mkdir -p gpurun_out/r03z
for D in 256 128; do
  timeout 300 python tools/lb_phase_timing.py build/variants/libzshmc_lbtiming.so $D 32768 50000
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z/lb_phase_timing_v2.txt
