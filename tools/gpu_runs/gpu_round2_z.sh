#!/bin/bash
mkdir -p gpurun_out/r02z gpurun_out/prof
timeout 1200 bash tools/profile.sh r02g > gpurun_out/r02z/profile.log 2>&1
sed -n 1,4p gpurun_out/prof/r02g_summary.txt | cut -c1-200
grep "per-launch\|SAME traced\|FETCH_SIZE\|WRITE_SIZE\|SQ_INSTS_VALU " gpurun_out/prof/r02g_summary.txt | cut -c1-330
