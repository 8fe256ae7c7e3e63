#!/bin/bash
mkdir -p gpurun_out/r02i
O=gpurun_out/r02i
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_base.so > $O/kbench3.txt 2>&1
timeout 300 python tools/kbench.py build/variants/libzshmc_base.so --adapt > $O/kbench3a.txt 2>&1
grep -hv amdgpu $O/kbench3.txt $O/kbench3a.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -6 $O/pytest.txt
timeout 600 python bench.py --steps 200 --warmup 20 --no-extra-configs --no-cpu-baseline --no-ess > $O/bench.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_adaptation_mode'])
PY
