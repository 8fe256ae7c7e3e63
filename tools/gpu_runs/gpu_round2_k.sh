#!/bin/bash
mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
KB_REPS=4 timeout 300 python tools/kbench.py build/variants/libzshmc_base.so build/variants/libzshmc_k1.so build/variants/libzshmc_k1p1.so build/variants/libzshmc_k1p2.so build/variants/libzshmc_k2p2.so > $O/kbench_dma.txt 2>&1
grep -v amdgpu $O/kbench_dma.txt
