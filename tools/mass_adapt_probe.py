#!/usr/bin/env python
"""Where the cost of a mass-adapting fused transition goes: host time per
`sample_op.run` call (measured on a tiny problem, where the device is idle),
wall time per transition at the headline shape, with both adaptation flags
on / off.    python tools/mass_adapt_probe.py [n_chains] [n_data]
PROBE_LIB = another build of libzshmc.so (tools/build_ring_variants.sh)."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402
from zhusuan_amd import _capi  # noqa: E402

if os.environ.get('PROBE_LIB'):   # another build (tools/build_variants.sh)
    _capi.LIB_PATH = os.path.abspath(os.environ['PROBE_LIB'])
    print('# library: %s' % _capi.LIB_PATH, flush=True)

dev = torch.device('cuda', 0)


def build(C, D, L=10):
    logstd = torch.linspace(-1, 1, D, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(D, device=dev), logstd=logstd,
                  n_samples=C, group_ndims=1)
        return bn
    x = torch.zeros(C, D, device=dev)
    f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
    hmc = zs.HMC(step_size=0.1, n_leapfrogs=L, adapt_step_size=f_ss,
                 adapt_mass=f_m, target_acceptance_rate=0.8, seed=3)
    op, info = hmc.sample(model(), {}, {'x': x})
    return hmc, op, f_ss, f_m


def loop(op, feed, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        op.run(feed_dict=feed, sync=False)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_host / n * 1e6, (time.perf_counter() - t0) / n * 1e6


C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gc.disable()
for shape in ((C, D),) if os.environ.get('PROBE_BIG_ONLY') else ((256, D), (C, D)):
    hmc, op, f_ss, f_m = build(*shape)
    for _ in range(40):
        op.run(feed_dict={f_ss: True, f_m: True}, sync=False)
    hmc.check_numerics()
    for label, on_ss, on_m in (('both flags on ', True, True),
                               ('step size only', True, False),
                               ('both flags off', False, False)):
        feed = {f_ss: on_ss, f_m: on_m}
        loop(op, feed, 60)
        host, wall = loop(op, feed, 300)
        print('%6d x %d  %s  host %.1f us per call, wall %.1f us per transition'
              % (shape[0], shape[1], label, host, wall), flush=True)
    del hmc, op
