#!/bin/bash
# round 3, first GPU trip: the whole GPU suite on the refactored host flow
# (end-of-run all-reduce, symbolic spellings, sharded non-fused plans) and the
# new bench line
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 25 gpurun_out/r03a/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r03a/bench.err; cut -c1-600 gpurun_out/r03a/bench.json
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03a/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'frac', d['roofline']['frac'])
    for e in d.get('extra_configs', []):
        print({k: e.get(k) for k in ('plan', 'ms_per_step', 'mean_acceptance', 'step_size', 'error')},
              e.get('roofline', {}).get('frac'), e.get('ess', {}).get('ess_per_sec'))
except Exception as e:
    print('parse failed', e)
PY
