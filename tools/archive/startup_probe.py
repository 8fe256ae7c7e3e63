#!/usr/bin/env python
"""What does the first fused launch after a synchronisation cost?  Event
time over N back-to-back launches issued right after torch.cuda.synchronize()
(N = 1 .. 40), plain and with an idle gap of 0 / 1 / 20 ms before them."""
import sys
import time

import torch

sys.path.insert(0, '.')
import zhusuan_amd as zs  # noqa: E402

dev = torch.device('cuda', 0)
C, D, L = 65536, 1024, 10
logstd = torch.linspace(-1.0, 1.0, D, device=dev)
mean = torch.zeros(D, device=dev)


@zs.meta_bayesian_net()
def gaussian():
    bn = zs.BayesianNet()
    bn.normal('x', mean, logstd=logstd, n_samples=C, group_ndims=1)
    return bn


x = torch.zeros(C, D, device=dev)
hmc = zs.HMC(step_size=0.14, n_leapfrogs=L, seed=1)
op, info = hmc.sample(gaussian(), {}, {'x': x})
plan = hmc._plan
plan.collect_acc = False
stream = torch.cuda.current_stream().cuda_stream
for i in range(300):
    plan._launch(i + 1, None, 1, L, stream)
torch.cuda.synchronize()
t = 400
for gap_ms in (0.0, 1.0, 20.0):
    row = []
    for n in (1, 2, 3, 5, 10, 20, 40):
        best = 1e9
        for rep in range(3):
            for i in range(50):                      # busy again
                plan._launch(t, None, 1, L, stream); t += 1
            torch.cuda.synchronize()
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                plan._launch(t, None, 1, L, stream); t += 1
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append('N=%d: %.1f us total, %.1f/launch' % (n, best * 1e3, best * 1e3 / n))
    print('idle gap %4.1f ms | ' % gap_ms + ' | '.join(row))
