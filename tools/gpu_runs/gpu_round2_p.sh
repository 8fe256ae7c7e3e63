#!/bin/bash
mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
