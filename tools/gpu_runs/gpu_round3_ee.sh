#!/bin/bash
# round 3: the last product change (D = 256 phase-3b read-ahead) -- full suite
mkdir -p gpurun_out/r03ee
( time timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r03ee/pytest_gpu.log
timeout 120 python tools/lb_wide_bench.py 32768 65536 2>&1 | grep "D=256" | cut -c1-170 | tee gpurun_out/r03ee/d256.txt
