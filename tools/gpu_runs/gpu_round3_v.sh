#!/bin/bash
# round 3: A/B of the feature-split decomposition at D = 256 (two workgroups
# per CU) against the 64-chain-block kernel
mkdir -p gpurun_out/r03v
ZSHMC_LB_SPLIT256=1 timeout 600 python -m pytest tests/test_gpu_linear_bernoulli.py -x -q -k "float64_reference or row_range" 2>&1 | tail -3
for c in 16384 32768; do
  echo "== 64-chain-block kernel, C=$c"; timeout 300 python tools/lb_wide_bench.py $c 65536 2>&1 | grep "D=256"
  echo "== feature-split kernel, C=$c"; ZSHMC_LB_SPLIT256=1 timeout 300 python tools/lb_wide_bench.py $c 65536 2>&1 | grep "D=256"
done 2>&1 | tee gpurun_out/r03v/split256_ab.txt
