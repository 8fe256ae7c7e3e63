#!/bin/bash
mkdir -p gpurun_out/r02final
O=gpurun_out/r02final
for rep in 1 2; do
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_$rep.txt 2>&1
echo "pytest run $rep exit $?" | tee -a $O/pytest_gpu_$rep.txt
tail -2 $O/pytest_gpu_$rep.txt | head -1
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench (no flags) wall ${SECONDS}s rc=$?"
python - <<PY
import json
d=json.load(open('$O/bench.json'))
r=d['roofline']
print('value %.4g ms/step %.4f steps %d | kernel_ms %.4f frac %.3f | other %s' % (d['value'], d['ms_per_step'], d['steps'], r['kernel_ms'], r['frac'], d['other_adaptation_mode']))
print(d.get('ess'))
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('cpu_')})
for e in d.get('extra_configs', []): print(e.get('workload','')[:40], e.get('ms_per_step'), e.get('mean_acceptance_first_transition'), e.get('roofline',{}).get('frac'), e.get('roofline',{}).get('sustained_over_transition'), e.get('error'))
PY
