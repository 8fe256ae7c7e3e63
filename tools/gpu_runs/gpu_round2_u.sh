#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r02u/pytest.txt
timeout 300 python tools/generic_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
