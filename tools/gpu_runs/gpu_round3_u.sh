#!/bin/bash
# round 3: full GPU suite after the packed-state / wide-kernel work
mkdir -p gpurun_out/r03u
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/r03u/pytest_gpu.log
