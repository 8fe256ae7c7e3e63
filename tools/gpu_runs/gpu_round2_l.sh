#!/bin/bash
mkdir -p gpurun_out/r02l gpurun_out/prof
O=gpurun_out/r02l
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
# two ranks on one GPU with the DEFAULT backend: RCCL refuses -> watchdog -> gloo fallback
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 2 --chains-per-gpu 4096 --no-ess > $O/bench2.json 2> $O/bench2.err
tail -2 $O/bench2.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open('$O/bench2.json') if l.startswith('{')][-1]
    print('2-rank:', d['n_gpus'], d['rccl_ranks'], d['collective'][:120], d['value'])
except Exception as e: print('2-rank bench failed', e)
PY
timeout 900 bash tools/profile_native.sh r02d 100000 128 > $O/profile_native.log 2>&1; tail -4 $O/profile_native.log | cut -c1-200
timeout 900 bash tools/profile.sh r02b > $O/profile.log 2>&1; grep -A3 "kernel stats" $O/profile.log | cut -c1-200
