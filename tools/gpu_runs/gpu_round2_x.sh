#!/bin/bash
mkdir -p gpurun_out/r02x
{
SECONDS=0
timeout 900 python examples/topic_model_mcem.py --epochs 3 2>&1 | grep -v amdgpu.ids | tail -12
echo "topic model full size: ${SECONDS}s"
SECONDS=0
timeout 600 python examples/logistic_regression_hmc.py 2>&1 | grep -v amdgpu.ids | tail -8
echo "logistic regression default: ${SECONDS}s"
} | tee gpurun_out/r02x/examples.txt
