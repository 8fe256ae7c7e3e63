#!/usr/bin/env python
"""BASELINE config 5 end to end: the LNTM E-step sampled by HMC at 8 192
(chain, document) rows x K = 128 topics x V = 12 419 words, L = 20, step size
and mass adaptation on (generic plan: autograd glue around the fused
mixture-multinomial kernel).  Prints time per transition and the sustained
TFLOP/s of the likelihood + gradient evaluations (4*R*K*V flop each)."""
import sys
import time
import torch
sys.path.insert(0, '.')
import zhusuan_amd as zs  # noqa: E402

CH, DOCS, K, V, L = 2, 4096, 128, 12419, 20
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
beta = torch.randn(K, V, device=dev, generator=g)
phi = torch.softmax(beta, -1)
x = torch.poisson(torch.full((DOCS, V), 0.08, device=dev), generator=g)
eta_mean = torch.zeros(DOCS, K, device=dev)
eta_logstd = torch.zeros(K, device=dev)


@zs.meta_bayesian_net()
def lntm():
    bn = zs.BayesianNet()
    eta = bn.normal('eta', eta_mean, logstd=eta_logstd, n_samples=CH,
                    group_ndims=1)
    bn.unnormalized_multinomial(
        'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
        normalize_logits=False, dtype=torch.float32)
    return bn


zs.set_random_seed(3)
hmc = zs.HMC(step_size=1e-3, n_leapfrogs=L, adapt_step_size=True,
             adapt_mass=True, target_acceptance_rate=0.6)
eta = torch.zeros(CH, DOCS, K, device=dev)
op, info = hmc.sample(lntm(), {'x': x}, {'eta': eta})
for _ in range(3):
    op.run()
torch.cuda.synchronize()
n = 10
t0 = time.time()
for _ in range(n):
    op.run(sync=False)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
flop = 4.0 * CH * DOCS * K * V * (L + 1)
print('config 5: %d rows x K=%d x V=%d, L=%d: %.1f ms per transition, '
      '%.1f TFLOP/s sustained (%.1f%% of 157.3), acc %.3f, step %.4f' % (
          CH * DOCS, K, V, L, dt * 1e3, flop / dt / 1e12,
          flop / dt / 1e12 / 1.573, float(info.acceptance_rate.mean()),
          float(info.updated_step_size)))
