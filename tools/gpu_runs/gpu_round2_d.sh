#!/bin/bash
# GPU call D of round 2: native plans (tests), bench with extra configs at
# reduced config-5 size, rocprofv3 profiles of headline and native plans.
mkdir -p gpurun_out/r02d gpurun_out/prof
O=gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py tests/test_gpu_fused.py tests/test_gpu_two_rank.py -m gpu -q > $O/pytest_models.txt 2>&1
echo "pytest exit $?" >> $O/pytest_models.txt
tail -6 $O/pytest_models.txt
timeout 900 python bench.py --steps 200 --warmup 20 --config5-chains 256 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
for e in d.get('extra_configs', []): print(json.dumps(e)[:900])
PY
timeout 900 bash tools/profile.sh r02a > $O/profile.log 2>&1; tail -30 $O/profile.log
timeout 900 bash tools/profile_native.sh r02a 100000 128 > $O/profile_native.log 2>&1; tail -45 $O/profile_native.log
