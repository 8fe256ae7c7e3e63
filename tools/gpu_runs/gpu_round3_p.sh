#!/bin/bash
# round 3: on top of nt row stores -- later DMA issue, sc0 sc1 stores, sc0 sc1
# loads, ring depth 2
mkdir -p gpurun_out/r03p
{
for opt in "" "--mass" "--adapt"; do
  echo "== kbench $opt"
  KB_REPS=6 timeout 900 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_d1.so build/variants/libzshmc_d2.so build/variants/libzshmc_st2.so build/variants/libzshmc_ld2.so build/variants/libzshmc_k2.so build/variants/libzshmc_base2.so $opt
done
} > gpurun_out/r03p/kbench.txt 2>&1
grep "==\|best" gpurun_out/r03p/kbench.txt | cut -c1-200
