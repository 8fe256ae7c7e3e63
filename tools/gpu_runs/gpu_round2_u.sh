#!/bin/bash
mkdir -p gpurun_out/r02final
O=gpurun_out/r02final
SECONDS=0
timeout 1500 python bench.py > $O/bench2.json 2> $O/bench2.err
echo "bench (no flags) wall ${SECONDS}s rc=$?"
python - <<PY
import json
d=json.load(open('$O/bench2.json'))
r=d['roofline']
print('value %.4g ms/step %.4f steps %d | kernel_ms %.4f frac %.3f | other %s' % (d['value'], d['ms_per_step'], d['steps'], r['kernel_ms'], r['frac'], d['other_adaptation_mode']))
for e in d.get('extra_configs', []): print(e.get('workload','')[:40], e.get('ms_per_step'), e.get('mean_acceptance_first_transition'), e.get('roofline',{}).get('frac'), e.get('roofline',{}).get('sustained_over_transition'), e.get('error'))
PY
