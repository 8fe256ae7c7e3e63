#!/usr/bin/env python
"""Many-chain HMC on a diagonal Gaussian (BASELINE config 1, the workload of
the reference's toy Gaussian example): target N(0, diag(1/(j+1))^2), 1 000
chains, L = 5, target acceptance 0.9, step size and mass adapted during the
first quarter of the run through per-run feedable flags.

Shows the drop-in surface: a `@zs.meta_bayesian_net` model, `zs.HMC(...)
.sample(model, observed, latent)`, `zs.Session().run(fetches, feed_dict)`.
The sampler recognises the single diag-Normal node and runs the whole
transition in one fused kernel.

    python examples/diag_gaussian_hmc.py [--dim 10] [--chains 1000] [--iters 200]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402


def make_target(scales, n_chains):
    """x ~ N(0, diag(scales^2)), one row of x per chain."""
    @zs.meta_bayesian_net()
    def target():
        net = zs.BayesianNet()
        net.normal('x', torch.zeros_like(scales), std=scales,
                   n_samples=n_chains, group_ndims=1)
        return net
    return target()


def sample(args, device):
    scales = 1.0 / torch.arange(1, args.dim + 1, dtype=torch.float32,
                                device=device)
    warm = zs.placeholder(bool, name='warm')      # adapt while True
    sampler = zs.HMC(step_size=1e-3, n_leapfrogs=5, adapt_step_size=warm,
                     adapt_mass=warm, target_acceptance_rate=0.9)
    state = torch.zeros(args.chains, args.dim, device=device)
    step, info = sampler.sample(make_target(scales, args.chains), {},
                                {'x': state})
    print('sampler plan:', sampler.plan_kind)
    fetches = [step, info.samples['x'], info.acceptance_rate,
               info.updated_step_size]
    kept = []
    n_warm = args.iters // 4
    with zs.Session() as sess:
        for it in range(args.iters):
            _, x, acc, eps = sess.run(fetches, feed_dict={warm: it < n_warm})
            if it < 12 or it % 25 == 0:
                print('  it %3d  acceptance %.3f  next step size %.5f' %
                      (it, acc.mean(), eps))
            if it >= args.iters // 2:
                kept.append(x)
    return np.concatenate(kept), scales.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dim', type=int, default=10)
    ap.add_argument('--chains', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=200)
    args = ap.parse_args()
    zs.set_random_seed(1)
    draws, scales = sample(args, torch.device('cuda', 0))
    rel = draws.std(0) / scales - 1.0
    print('%d draws; |mean| max %.4f; std relative error min %+.4f max %+.4f' %
          (len(draws), np.abs(draws.mean(0)).max(), rel.min(), rel.max()))
    print('Relative error of stdev = {}'.format(np.round(rel, 4)))
    assert np.abs(draws.mean(0)).max() < 0.05 and np.abs(rel).max() < 0.05


if __name__ == '__main__':
    main()
