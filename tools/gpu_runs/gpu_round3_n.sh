#!/bin/bash
# round 3: cache-policy combinations on the row traffic of the ring kernel
# (the default library first AND last: position in the rotation matters while
# the clocks settle)
mkdir -p gpurun_out/r03n
{
for opt in "" "--mass" "--mean"; do
  echo "== kbench $opt"
  KB_REPS=6 timeout 900 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_ntnt.so build/variants/libzshmc_ntntd1.so build/variants/libzshmc_st3.so build/variants/libzshmc_st2.so build/variants/libzshmc_ld3.so build/variants/libzshmc_stnt.so build/variants/libzshmc_base2.so $opt
done
} > gpurun_out/r03n/kbench_policy.txt 2>&1
grep "==\|best" gpurun_out/r03n/kbench_policy.txt | cut -c1-200
