#!/bin/bash
# round 3: the bench line as the driver runs it, with the new extra
mkdir -p gpurun_out/r03w
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03w/bench_driver_flags.json 2> gpurun_out/r03w/bench_driver_flags.err
tail -c 600 gpurun_out/r03w/bench_driver_flags.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03w/bench_driver_flags.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for e in d.get('extra_configs', []):
    print(e.get('workload', '')[:60], '|', e.get('plan'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), e.get('mean_acceptance'), e.get('error'))
PY
