#!/bin/bash
# GPU call E of round 2: full GPU suite with native plans + AIS pin, bench with
# reduced config-5 size, native-plan profile.
mkdir -p gpurun_out/r02e gpurun_out/prof
O=gpurun_out/r02e
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
timeout 900 bash tools/profile_native.sh r02b 100000 128 > $O/profile_native.log 2>&1; tail -60 $O/profile_native.log | cut -c1-180
timeout 900 python bench.py --steps 200 --warmup 20 --config5-chains 256 --no-ess > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
for e in d.get('extra_configs', []): print(json.dumps(e)[:1200])
PY
