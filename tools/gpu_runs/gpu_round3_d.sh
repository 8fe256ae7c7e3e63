#!/bin/bash
# round 3, fourth trip: GPU suite after the likelihood's two-level sum and the
# LDS-atomic column statistics; bench
mkdir -p gpurun_out/r03d
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03d/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 60 gpurun_out/r03d/pytest.log | cut -c1-400
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r03d/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03d/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'])
    print('other', d['other_adaptation_mode'])
    print('mass', json.dumps(d['mass_adaptation_modes']))
    for e in d.get('extra_configs', []):
        print({k: e.get(k) for k in ('plan', 'ms_per_step', 'mean_acceptance', 'mean_acceptance_subset_held_phase', 'step_size', 'error')},
              e.get('roofline', {}).get('frac'), e.get('ess'))
except Exception as e:
    print('parse failed', e)
PY
