#!/bin/bash
# GPU call A of round 2: instruction costs, A/B of the fused kernel variants,
# the GPU test suite, a bench line.
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
rocminfo | grep -m3 "Marketing Name\|Compute Unit" > $O/rocminfo.txt 2>&1
nproc > $O/nproc.txt
timeout 120 ./build/instr_bench > $O/instr_bench.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so build/variants/libzshmc_nobitop3.so build/variants/libzshmc_epsv.so > $O/kbench_zero_mean.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so build/variants/libzshmc_epsv.so --mean > $O/kbench_mean.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_base.so --adapt > $O/kbench_adapt.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so --mass > $O/kbench_mass.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest_gpu.txt; cat $O/kbench_zero_mean.txt; tail -3 $O/bench.err; head -c 600 $O/bench.json
