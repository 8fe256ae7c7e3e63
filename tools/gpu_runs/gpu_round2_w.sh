#!/bin/bash
mkdir -p gpurun_out/r02w
O=gpurun_out/r02w
for K in 20 20 200 10; do
timeout 600 python bench.py --gpus 1 --steps $K --warmup 5 --no-extra-configs --no-cpu-baseline --no-ess > $O/bench$K.json 2> $O/bench$K.err
python - <<PY
import json
d=json.load(open('$O/bench$K.json'))
r=d['roofline']
print($K, 'value %.4g ms/step %.4f | kernel_ms %.4f region %.4f b2b %.4f frac %.3f n=%d' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_timed_region'], r['kernel_ms_back_to_back'], r['frac'], r['kernel_launches_timed']))
PY
done
