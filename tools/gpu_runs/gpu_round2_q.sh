#!/bin/bash
mkdir -p gpurun_out/r02q gpurun_out/prof
O=gpurun_out/r02q
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 900 bash tools/profile_native.sh r02f 100000 128 > $O/profile_native.log 2>&1
tail -3 $O/profile_native.log
grep -B2 -A16 "kernels launched inside" gpurun_out/prof/r02f_native_summary.txt | cut -c1-150
SECONDS=0
timeout 1500 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench wall ${SECONDS}s" | tee -a $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
for e in d.get('extra_configs', []): print(json.dumps(e)[:1500])
PY
