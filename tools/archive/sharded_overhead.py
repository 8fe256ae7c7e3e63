#!/usr/bin/env python
"""What the sharded-chain path adds to an adaptive transition on ONE GPU:
the same 65 536 x 1 024 workload run (a) unsharded, (b) through a one-rank
RCCL communicator with always_reduce (ncclAllReduce of the 2 statistics
words + the update applied in the next launch's prologue).  Prints GPU time
per step and the host's enqueue time per step (the loop is GPU-bound only
while the second stays below the first)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, '.')
import zhusuan_amd as zs  # noqa: E402
from zhusuan_amd.distributed import ChainSharding  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('gloo', rank=0, world_size=1)
dev = torch.device('cuda', 0)
C, D, L = 65536, 1024, 10
logstd = torch.linspace(-1.0, 1.0, D, device=dev)
mean = torch.zeros(D, device=dev)


def run(sharding, label, n=400):
    @zs.meta_bayesian_net()
    def gaussian():
        bn = zs.BayesianNet()
        bn.normal('x', mean, logstd=logstd, n_samples=C, group_ndims=1)
        return bn
    x = torch.zeros(C, D, device=dev)
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=True,
                 target_acceptance_rate=0.8, seed=1, sharding=sharding)
    op, info = hmc.sample(gaussian(), {}, {'x': x})
    for _ in range(120):
        op.run(sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        op.run(sync=False)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('%-34s %.4f ms/step on the GPU, host enqueue %.4f ms/step, eps %.5f'
          % (label, t_all / n * 1e3, t_enq / n * 1e3,
             float(info.updated_step_size.item())))


run(None, 'unsharded, adaptive (first in process)')
run(None, 'unsharded, adaptive')
sh = ChainSharding(backend='rccl', always_reduce=True, chain_offset=0,
                   n_chains_global=C)
run(sh, '1-rank RCCL all-reduce, adaptive')
sh.close()
sh = ChainSharding(backend='torch', always_reduce=True, chain_offset=0,
                   n_chains_global=C)
run(sh, 'torch backend (no-op at 1 rank)')
run(None, 'unsharded, adaptive (again)')
