#!/usr/bin/env python
"""Why are the first few hundred sample_op.run calls of a process slow on the
host?  Per-20-run averages of: the whole run, the model re-evaluation
(refresh_model) and the C call that launches the kernel.
argv: [adapt=0|1] [n=600]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import zhusuan_amd as zs  # noqa: E402
from zhusuan_amd import _capi  # noqa: E402

args = dict(a.split('=') for a in sys.argv[1:])
adapt_on = args.get('adapt', '1') == '1'
n = int(args.get('n', 600))
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
flag = zs.placeholder(bool)
hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, adapt_step_size=flag,
             target_acceptance_rate=0.8, seed=1)
op, info = hmc.sample(gaussian(), {}, {'x': x})
plan = hmc._plan
acc = {'refresh': 0.0, 'call': 0.0}
orig_refresh, orig_call = plan.refresh_model, _capi.call


def refresh():
    t = time.perf_counter()
    orig_refresh()
    acc['refresh'] += time.perf_counter() - t


def call(*a):
    t = time.perf_counter()
    r = orig_call(*a)
    acc['call'] += time.perf_counter() - t
    return r


plan.refresh_model = refresh
_capi.call = call
import zhusuan_amd.hmc as H  # noqa: E402
H._capi.call = call
rows = []
for i in range(n):
    if i % 20 == 0:
        t0 = time.perf_counter()
        acc['refresh'] = acc['call'] = 0.0
    op.run(feed_dict={flag: adapt_on or i < 3}, sync=False)
    if i % 20 == 19:
        dt = time.perf_counter() - t0
        rows.append('%4d: run %.3f  refresh %.3f  C call %.3f ms' % (
            i + 1, dt / 20 * 1e3, acc['refresh'] / 20 * 1e3,
            acc['call'] / 20 * 1e3))
torch.cuda.synchronize()
print('adapt', adapt_on)
print('\n'.join(rows))
