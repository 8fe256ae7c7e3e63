#!/bin/bash
# round 3, ninth trip: final-code validation -- GPU suite, the element-wise
# kernels' bandwidth table, the full default bench line
mkdir -p gpurun_out/r03i
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03i/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 gpurun_out/r03i/pytest.log | cut -c1-300
timeout 600 python tools/generic_bench.py > gpurun_out/r03i/generic_bench.txt 2>&1
tail -n 40 gpurun_out/r03i/generic_bench.txt | cut -c1-200
timeout 1500 python bench.py > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/r03i/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03i/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'], 'steps', d['steps'])
print('other', d['other_adaptation_mode']); print('pyloop', d['python_loop'])
print('mass', json.dumps(d['mass_adaptation_modes']))
for e in d.get('extra_configs', []):
    if 'run_many' in e:
        print({k: e[k] for k in ('run_many', 'python_loop')}); continue
    print({k: e.get(k) for k in ('plan', 'ms_per_step', 'mean_acceptance', 'mean_acceptance_subset_held_phase', 'step_size', 'error')},
          e.get('roofline', {}).get('frac'), e.get('ess', {}).get('ess_per_sec'))
PY
