#!/bin/bash
# round 3: final code -- rocprofv3 evidence of the headline command, the full
# default bench line
mkdir -p gpurun_out/r03q gpurun_out/prof
timeout 1500 bash tools/profile.sh r03q > gpurun_out/r03q/profile.log 2>&1
sed -n 1,14p gpurun_out/prof/r03q_summary.txt | cut -c1-260
grep "FETCH_SIZE\|WRITE_SIZE\|SQ_INSTS_VALU \|timed region\|SAME traced" gpurun_out/prof/r03q_summary.txt | cut -c1-300
timeout 1500 python bench.py > gpurun_out/r03q/bench.json 2> gpurun_out/r03q/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03q/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'], 'steps', d['steps'], 'ess', d['ess']['ess_per_sec'])
print('other', d['other_adaptation_mode']); print('pyloop', d['python_loop'])
print('mass', json.dumps(d['mass_adaptation_modes']))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for e in d.get('extra_configs', []):
    if 'run_many' in e:
        print({k: e[k] for k in ('run_many', 'python_loop')}); continue
    print({k: e.get(k) for k in ('plan', 'ms_per_step', 'mean_acceptance', 'mean_acceptance_subset_held_phase', 'step_size', 'error')},
          e.get('roofline', {}).get('frac'), e.get('ess', {}).get('ess_per_sec'))
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-ess > gpurun_out/r03q/bench20.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03q/bench20.json').read().strip().splitlines()[-1])
print('steps20 value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'])
PY
