#!/bin/bash
# round 3: final code -- full GPU suite, the default bench line, the line with
# the driver's flags, a kernel trace of the wide-kernel bench
mkdir -p gpurun_out/r03zz gpurun_out/prof
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r03zz/pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/r03zz/bench.json 2> gpurun_out/r03zz/bench.err
echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03zz/bench_driver_flags.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench', 'bench_driver_flags'):
    d = json.loads(open('gpurun_out/r03zz/%s.json' % f).read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'], 'steps', d['steps'])
    for e in d.get('extra_configs', []):
        print('   ', e.get('workload', '')[:44], '|', e.get('plan'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'),
              e.get('mean_acceptance'), (e.get('ess') or {}).get('ess_per_sec'), e.get('error'))
PY
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/r03zz_wide_trace -o trace --output-format csv -- python $R/tools/lb_wide_bench.py 8192 65536 > $R/gpurun_out/r03zz/wide_trace.log 2>&1
cd $R
f=$(find gpurun_out/prof/r03zz_wide_trace -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200 | tee gpurun_out/r03zz/wide_kernel_stats_head.txt
