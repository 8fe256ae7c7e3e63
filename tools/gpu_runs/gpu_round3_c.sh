#!/bin/bash
# round 3, third trip: A/B of the COLSTATS accumulation modes on the headline
# shape, the fixed tests, the new full-size native-plan parity tests
mkdir -p gpurun_out/r03c
{
for opt in "--mass" "--mass --colstats" "" "--colstats"; do
  echo "== kbench $opt"
  timeout 300 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_csk1.so build/variants/libzshmc_csf32.so build/variants/libzshmc_cslds.so $opt
done
} > gpurun_out/r03c/kbench.txt 2>&1
cat gpurun_out/r03c/kbench.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03c/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 40 gpurun_out/r03c/pytest.log | cut -c1-400
