#!/usr/bin/env python
"""The E-step of examples/topic_models/lntm_mcem.py:62-70,157-182 at its own
sizes (one chain, 100 documents per minibatch, K = 100 topics, V = 12 419,
L = 20): wall time per transition on the native mixture-multinomial plan.
    python tools/estep_bench.py [one_launch 0|1] [n_transitions] [auto|fp32|bf16x3]
(ZSHMC_MIN_SLICE_ROWS=64|96|128|256: vocabulary rows per slice, A/B)
Under rocprofv3 --kernel-trace --stats: where a transition's time goes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402

one = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
arith = sys.argv[3] if len(sys.argv) > 3 else 'auto'
dev = torch.device('cuda', 0)
n_chains, n_docs, K, V, L = 1, 100, 100, 12419, 20
g = torch.Generator(device=dev).manual_seed(3)
phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
mean = torch.zeros(n_docs, K, device=dev)
logstd = torch.zeros(K, device=dev)


@zs.meta_bayesian_net()
def lntm():
    bn = zs.BayesianNet()
    eta = bn.normal('eta', mean, logstd=logstd, n_samples=n_chains,
                    group_ndims=1)
    theta = torch.softmax(eta.tensor, -1)
    bn.unnormalized_multinomial(
        'x', torch.log((theta.reshape(-1, K) @ phi).reshape(
            n_chains, n_docs, V)), normalize_logits=False, dtype=torch.float32)
    return bn


m = lntm()
m.log_joint = lambda bn: bn.cond_log_prob('eta') + bn.cond_log_prob('x')
hmc = zs.HMC(step_size=0.05, n_leapfrogs=L, seed=5, one_launch_trajectory=one,
             likelihood_arithmetic=arith)
eta = torch.zeros(n_chains, n_docs, K, device=dev)
op, info = hmc.sample(m, {'x': x}, {'eta': eta})
op.run_many(5)
torch.cuda.synchronize()
t0 = time.perf_counter()
op.run_many(n)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print('E-step transition (%s, %s, %d row-range slices of %d chain blocks): %.3f ms, '
      'mean acceptance %.3f' % (hmc.likelihood_arithmetic_used,
          'one launch' if one else 'launch per trip', hmc._plan.splits,
          (hmc._plan.lik_rows + hmc._plan.block - 1) // hmc._plan.block, ms,
          float(info.acceptance_rate.mean())))
