#!/bin/bash
# round 3, eighth trip: suite (examples on the literal spellings), rocprofv3
# evidence of the headline command (kernel trace + separate PMC passes)
mkdir -p gpurun_out/r03h gpurun_out/prof
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03h/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 gpurun_out/r03h/pytest.log | cut -c1-300
timeout 1500 bash tools/profile.sh r03h > gpurun_out/r03h/profile.log 2>&1
sed -n 1,40p gpurun_out/prof/r03h_summary.txt | cut -c1-260
