#!/bin/bash
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout 900 python -m pytest tests/test_gpu_mixture_multinomial.py tests/test_gpu_examples.py tests/test_gpu_lntm_ais.py tests/test_gpu_linear_bernoulli.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -8 $O/pytest.txt
timeout 600 python tools/native_plan_trace.py 100000 128 > $O/native.txt 2>&1; tail -2 $O/native.txt
timeout 1500 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
for e in d.get('extra_configs', []): print(json.dumps(e)[:1500])
PY
