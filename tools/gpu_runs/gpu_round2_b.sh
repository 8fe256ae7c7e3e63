#!/bin/bash
# GPU call B of round 2: collect-path cost after dropping the fences, tests.
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so > $O/kbench_zero_mean.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_base.so --adapt > $O/kbench_adapt.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so --mean > $O/kbench_mean.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -8 $O/pytest_gpu.txt; cat $O/kbench_zero_mean.txt $O/kbench_adapt.txt | grep -v amdgpu; tail -3 $O/bench.err; head -c 300 $O/bench.json
