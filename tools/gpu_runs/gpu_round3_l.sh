#!/bin/bash
# round 3, twelfth trip: hipGraph replay inside zshmc_hmc_diag_normal_run
mkdir -p gpurun_out/r03l
{
ZSHMC_RUN_GRAPH=0 timeout 300 python tools/run_graph_bench.py
ZSHMC_RUN_GRAPH=1 timeout 300 python tools/run_graph_bench.py
} > gpurun_out/r03l/run_graph.txt 2>&1
grep ZSHMC gpurun_out/r03l/run_graph.txt
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03l/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 30 gpurun_out/r03l/pytest.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-ess > gpurun_out/r03l/bench20.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03l/bench20.json').read().strip().splitlines()[-1])
print('steps20 value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'])
PY
