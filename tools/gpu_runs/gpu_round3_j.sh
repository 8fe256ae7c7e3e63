#!/bin/bash
# round 3, tenth trip: suite after the last test changes; a measured proxy for
# the workgroup-per-chain design: the SAME 256 MiB of state as 65 536 x 1 024
# cut into 2x / 4x as many chains of 512 / 256 latents (one wave per chain
# still: 2 / 1 chunks per lane, trips 2x / 4x shorter, i.e. the read -> write
# distance the design is after, WITHOUT the cross-wave exchange it would add)
mkdir -p gpurun_out/r03j
{
for cd in "65536 1024" "131072 512" "262144 256"; do
  set -- $cd
  echo "== C=$1 D=$2 (same bytes)"
  KB_C=$1 KB_D=$2 KB_REPS=4 timeout 300 python tools/kbench.py zhusuan_amd/lib/libzshmc.so
  KB_C=$1 KB_D=$2 KB_REPS=4 timeout 300 python tools/kbench.py zhusuan_amd/lib/libzshmc.so --mass
done
} > gpurun_out/r03j/kbench_chain_split.txt 2>&1
grep "==\|best" gpurun_out/r03j/kbench_chain_split.txt | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03j/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 gpurun_out/r03j/pytest.log | cut -c1-300
