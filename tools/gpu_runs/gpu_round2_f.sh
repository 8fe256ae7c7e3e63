#!/bin/bash
mkdir -p gpurun_out/r02f
O=gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_mixture_multinomial.py tests/test_gpu_examples.py tests/test_gpu_lntm_ais.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -8 $O/pytest.txt
timeout 900 bash tools/profile_native.sh r02c 100000 128 > $O/profile_native.log 2>&1; grep -A28 "native plan 1" $O/profile_native.log | cut -c1-170; tail -3 $O/profile_native.log
timeout 900 python bench.py --steps 200 --warmup 20 --config5-chains 256 --no-ess --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
for e in d.get('extra_configs', []): print(json.dumps(e)[:1200])
PY
