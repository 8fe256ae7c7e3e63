#!/bin/bash
# round 3, thirteenth trip: the ring kernel's A/B knobs again, now that the
# generator is 30 % cheaper (the balance between the VALU and memory bounds
# moved): ring depth 2, later DMA issue, nt loads / stores, 3 waves per SIMD
mkdir -p gpurun_out/r03m
{
for opt in "" "--mass"; do
  echo "== kbench $opt"
  KB_REPS=5 timeout 600 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_k2.so build/variants/libzshmc_dma1.so build/variants/libzshmc_dma2.so build/variants/libzshmc_ldnt.so build/variants/libzshmc_stnt.so build/variants/libzshmc_w3.so $opt
done
} > gpurun_out/r03m/kbench_knobs.txt 2>&1
grep "==\|best" gpurun_out/r03m/kbench_knobs.txt | cut -c1-190
