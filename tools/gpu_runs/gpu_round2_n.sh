#!/bin/bash
mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
timeout 600 python -m pytest tests/test_gpu_mixture_multinomial.py tests/test_gpu_lntm_ais.py -m gpu -q 2>&1 | tail -3
timeout 600 python tools/native_plan_trace.py 100000 128 > $O/native.txt 2>&1; tail -2 $O/native.txt
timeout 600 python tools/lntm_bench.py > $O/lntm_bench.txt 2>&1; tail -3 $O/lntm_bench.txt
