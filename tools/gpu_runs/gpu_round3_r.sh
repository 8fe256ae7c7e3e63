#!/bin/bash
# round 3: kernel trace of the native plans with the literal spellings -- which
# kernels run inside marked transitions (none from ATen)
mkdir -p gpurun_out/r03r gpurun_out/prof
timeout 900 bash tools/profile_native.sh r03r 100000 128 > gpurun_out/r03r/profile_native.log 2>&1
grep -v "^  void\|^  zshmc\|^  __amd" gpurun_out/prof/r03r_native_summary.txt | cut -c1-200 | head -40
grep -A12 "inside 3 transitions" gpurun_out/prof/r03r_native_summary.txt | cut -c1-170 | head -60
