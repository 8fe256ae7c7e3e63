#!/bin/bash
mkdir -p gpurun_out/r02o gpurun_out/prof
O=gpurun_out/r02o
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 1500 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['other_adaptation_mode'])
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('cpu_')})
for e in d.get('extra_configs', []): print(json.dumps(e)[:1300])
PY
timeout 900 bash tools/profile.sh r02e > $O/profile.log 2>&1; grep -A3 "kernel stats\|per-launch" $O/profile.log | cut -c1-330
