#!/bin/bash
mkdir -p gpurun_out/r02v
{
for rep in 1 2; do
echo "== product D=128"; timeout 300 python tools/lb_bench.py 32768 100000 1 128
echo "== 3 waves/SIMD (spills) D=128"; LB_LIB=build/variants/libzshmc_lb3.so timeout 300 python tools/lb_bench.py 32768 100000 1 128
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02v/lb3.txt
