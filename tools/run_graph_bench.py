#!/usr/bin/env python
"""zshmc_hmc_diag_normal_run at BASELINE configs[0]'s size (1 000 x 10-D,
L = 5): microseconds per transition with the stretches of 16 launches replayed
from a hipGraph (default) against the plain C-side launch loop
(ZSHMC_RUN_GRAPH=0; the switch is read once per process)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402

dev = torch.device('cuda', 0)
for C, D, L in ((1000, 10, 5), (4096, 64, 10), (65536, 1024, 10)):
    std = torch.exp(torch.linspace(-1, 1, D, device=dev))

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(D, device=dev), std=std, n_samples=C,
                  group_ndims=1)
        return bn
    x = torch.zeros(C, D, device=dev)
    hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, seed=1)
    op, info = hmc.sample(model(), {}, {'x': x})
    op.run_many(200, sync=False)
    torch.cuda.synchronize()
    n = 4000 if C * D < 10 ** 6 else 400
    t0 = time.perf_counter()
    op.run_many(n, sync=False)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print('ZSHMC_RUN_GRAPH=%s  %6d x %4d, L=%2d: %8.2f us per transition '
          '(mean acceptance %.3f)' % (os.environ.get('ZSHMC_RUN_GRAPH', '(on)'),
                                      C, D, L, us,
                                      float(info.acceptance_rate.mean())))
