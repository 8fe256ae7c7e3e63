#!/bin/bash
# round 3, fifth trip: suite + bench with run_many / configs[0]; Philox-7 A/B;
# full-shape PMC profile of the MFMA likelihood kernel; kernel trace of the
# mass-adapting loop
mkdir -p gpurun_out/r03e gpurun_out/prof
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/r03e/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 30 gpurun_out/r03e/pytest.log | cut -c1-300
{
for opt in "" "--mass"; do
  echo "== kbench $opt"
  KB_REPS=4 timeout 300 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_philox7.so $opt
done
} > gpurun_out/r03e/kbench_philox7.txt 2>&1
cat gpurun_out/r03e/kbench_philox7.txt | cut -c1-200
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
echo "bench rc=$?"; tail -c 800 gpurun_out/r03e/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03e/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'])
    print('other', d['other_adaptation_mode']); print('pyloop', d['python_loop'])
    print('mass', json.dumps(d['mass_adaptation_modes']))
    for e in d.get('extra_configs', []):
        if 'run_many' in e:
            print(json.dumps(e)[:900]); continue
        print({k: e.get(k) for k in ('plan', 'ms_per_step', 'mean_acceptance', 'mean_acceptance_subset_held_phase', 'step_size', 'error')},
              e.get('roofline', {}).get('frac'), e.get('ess'))
except Exception as e:
    print('parse failed', e)
PY
timeout 900 bash tools/profile_native_full.sh r03 > gpurun_out/r03e/profile_native_full.log 2>&1
cat gpurun_out/prof/r03_nativefull_summary.txt | cut -c1-220
