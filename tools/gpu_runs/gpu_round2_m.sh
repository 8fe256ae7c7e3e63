#!/bin/bash
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_fullsize.py tests/test_gpu_hmc_reference.py tests/test_gpu_two_rank.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 200 --warmup 20 --no-extra-configs --no-cpu-baseline --no-ess > $O/bench$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/bench$i.json'))
o=d['other_adaptation_mode']
print(d['ms_per_step'], d['roofline']['frac'], o['ms_per_step'], 'adaptive overhead %.2f%%' % (100*(o['ms_per_step']/d['ms_per_step']-1)))
PY
done
