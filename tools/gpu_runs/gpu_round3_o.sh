#!/bin/bash
# round 3: nt row stores adopted -- A/B against the previous default on the
# colstats / mass instantiations, the suite, the bench line
mkdir -p gpurun_out/r03o
{
for opt in "" "--mass" "--mass --colstats" "--colstats" "--adapt"; do
  echo "== kbench $opt"
  KB_REPS=5 timeout 600 python tools/kbench.py zhusuan_amd/lib/libzshmc.so build/variants/libzshmc_st0.so $opt
done
} > gpurun_out/r03o/kbench_stnt.txt 2>&1
grep "==\|best" gpurun_out/r03o/kbench_stnt.txt | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03o/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 5 gpurun_out/r03o/pytest.log | cut -c1-300
for st in 20 200; do
timeout 600 python bench.py --steps $st --warmup 5 --no-extra-configs --no-cpu-baseline --no-ess > gpurun_out/r03o/bench$st.json 2>/dev/null
python - <<PY
import json
d = json.loads(open('gpurun_out/r03o/bench$st.json').read().strip().splitlines()[-1])
print('steps$st value', d['value'], 'frac', d['roofline']['frac'], 'ms', d['ms_per_step'], 'mass', d['mass_adaptation_modes']['overhead_of_adapting'])
PY
done
