#!/bin/bash
# round 3: where a tile of the feature-split kernel spends its clocks
mkdir -p gpurun_out/r03z
for D in 1024 512; do
  timeout 300 python tools/lbw_phase_timing.py build/variants/libzshmc_lbwtiming.so $D 8192 16384
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03z/lbw_phase_timing.txt
