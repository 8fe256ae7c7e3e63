#!/bin/bash
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
for s in sghmc sgld sgnht; do
  echo "== $s"; timeout 600 python examples/bayesian_nn_sgmcmc.py --small --sampler $s 2>&1 | grep -v amdgpu.ids | tail -12
done > $O/bnn.txt 2>&1
cat $O/bnn.txt
timeout 600 python - > $O/c3.txt 2>&1 <<'PY'
import json, torch, sys
sys.path.insert(0, '.')
import bench, zhusuan_amd as zs
dev = torch.device('cuda', 0)
print(json.dumps(bench.extra_config3(torch, zs, dev))[:1200])
PY
grep -v amdgpu.ids $O/c3.txt | tail -3
