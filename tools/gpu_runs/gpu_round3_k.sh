#!/bin/bash
# round 3, eleventh trip: suite; the TWO-RANK bench lines on the one-GPU box
# (both ranks on cuda:0, collectives over gloo: RCCL refuses two ranks on one
# device) -- configs[4] as the line's own workload and the default line with
# the sharded configs[4] extra, launched the way the driver launches them
mkdir -p gpurun_out/r03k
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03k/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 gpurun_out/r03k/pytest.log | cut -c1-300
export ZSHMC_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --workload lntm --steps 2 --warmup 1 --lntm-chains-per-gpu 256 > gpurun_out/r03k/bench_lntm_2rank_gloo.json 2> gpurun_out/r03k/bench_lntm_2rank.err
echo "lntm 2-rank rc=$?"; tail -c 300 gpurun_out/r03k/bench_lntm_2rank.err; grep '^{' gpurun_out/r03k/bench_lntm_2rank_gloo.json | cut -c1-700
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 2 --steps 20 --warmup 5 --chains-per-gpu 32768 --lntm-chains-per-gpu 128 > gpurun_out/r03k/bench_2rank_gloo.json 2> gpurun_out/r03k/bench_2rank.err
echo "default 2-rank rc=$?"; tail -c 300 gpurun_out/r03k/bench_2rank.err; grep '^{' gpurun_out/r03k/bench_2rank_gloo.json | cut -c1-500
