#!/bin/bash
mkdir -p gpurun_out/r02u
SECONDS=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-ess > gpurun_out/r02u/bench2.json 2> gpurun_out/r02u/bench2.err
echo "rc=$? wall ${SECONDS}s"
tail -3 gpurun_out/r02u/bench2.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('gpurun_out/r02u/bench2.json').read().strip().split('\n')[-1])
print({k: d[k] for k in ('value','n_gpus','ms_per_step','rccl_ranks','collective','scaling')})
print(d['roofline']['kernel_timing'], d['roofline']['frac'])
PY
