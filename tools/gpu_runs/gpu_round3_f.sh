#!/bin/bash
# round 3, sixth trip: the GPU suite on the Philox4x32-7 stream (regenerated
# goldens), bench
mkdir -p gpurun_out/r03f
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03f/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 45 gpurun_out/r03f/pytest.log | cut -c1-300
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r03f/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03f/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'], 'acc', d['mean_acceptance'], 'ess', d.get('ess', {}).get('ess_per_sec'))
    print('other', d['other_adaptation_mode']); print('pyloop', d['python_loop'])
    print('mass', json.dumps(d['mass_adaptation_modes']))
    print('cpu', d.get('cpu_baseline'))
except Exception as e:
    print('parse failed', e)
PY
timeout 600 python bench.py --steps 200 --warmup 5 --no-extra-configs --no-cpu-baseline --no-ess > gpurun_out/r03f/bench200.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03f/bench200.json').read().strip().splitlines()[-1])
print('steps200 value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'])
PY
