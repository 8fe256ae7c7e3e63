#!/bin/bash
mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
timeout 300 python tools/kbench.py build/variants/libzshmc_base.so build/variants/libzshmc_k1.so build/variants/libzshmc_w2.so build/variants/libzshmc_w2k1.so build/variants/libzshmc_w3k1.so > $O/kbench_k.txt 2>&1
grep -v amdgpu $O/kbench_k.txt
