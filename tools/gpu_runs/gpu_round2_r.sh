#!/bin/bash
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
V=build/variants
KB_REPS=4 timeout 600 python tools/kbench.py $V/libzshmc_base.so $V/libzshmc_stag1.so $V/libzshmc_stag2.so $V/libzshmc_stag3.so $V/libzshmc_w2.so $V/libzshmc_w2s1.so $V/libzshmc_s1k2.so $V/libzshmc_w3s1.so > $O/kbench.txt 2>&1
cat $O/kbench.txt | grep -v amdgpu.ids
