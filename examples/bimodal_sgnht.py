#!/usr/bin/env python
"""Stochastic-gradient Nose-Hoover thermostat on a two-mode 1-D target (the
workload of the reference's toy SGNHT example): 0.5 N(-1, 0.5^2) +
0.5 N(3, 0.5^2), 1 000 chains, scalar friction, first-order integrator.

The log density is a plain torch function of the latent dict (its gradient
comes from autograd); the thermostat update runs in the HIP kernels behind
zshmc_sgnht_update / zshmc_sgnht_scalar.  Only every 100th post-burn-in state
is copied to the host; the other iterations are enqueued without a round trip.

    python examples/bimodal_sgnht.py [--iters 30000]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402

MODES = (-1.0, 3.0)
WIDTH = 0.5


def log_density(latent):
    x = latent['x']
    parts = torch.stack([-0.5 * ((x - m) / WIDTH) ** 2 for m in MODES])
    return torch.logsumexp(parts, dim=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30000)
    ap.add_argument('--chains', type=int, default=1000)
    args = ap.parse_args()
    zs.set_random_seed(1)
    torch.manual_seed(1)
    device = torch.device('cuda', 0)

    thermostat = zs.SGNHT(learning_rate=0.2, variance_extra=0.1,
                          tune_rate=0.01, second_order=False,
                          use_vector_alpha=False)
    x = torch.empty(args.chains, device=device).uniform_(-5.0, 5.0)
    step, info = thermostat.sample(log_density, observed={}, latent={'x': x})

    burn = args.iters * 2 // 3
    thinned = []
    sess = zs.Session()
    for t in range(args.iters):
        report = t % 500 == 0
        keep = t >= burn and t % 100 == 0
        if not (report or keep):
            step.run()
            continue
        _, snap = sess.run([step, info])
        if report:
            print('t %6d  kinetic mean %.4f  friction %.4f' %
                  (t, float(snap.mean_k['x']), float(snap.alpha['x'])))
        if keep:
            thinned.append(snap.q['x'])
    draws = np.concatenate(thinned)

    mean = 0.5 * sum(MODES)
    std = math.sqrt(WIDTH ** 2 + 0.5 * sum(m * m for m in MODES) - mean ** 2)
    rel = draws.std() / std - 1.0
    print('mean %.4f (exact %.1f)   std %.4f (exact %.4f)' %
          (draws.mean(), mean, draws.std(), std))
    print('Relative error of stdev = {}'.format(rel))
    hist, edges = np.histogram(draws, bins=40, range=(-4, 6), density=True)
    mid = 0.5 * (edges[1:] + edges[:-1])
    pdf = sum(0.5 * np.exp(-0.5 * ((mid - m) / WIDTH) ** 2) for m in MODES) / \
        (WIDTH * math.sqrt(2 * math.pi))
    print('max |histogram - density| = %.3f' % np.abs(hist - pdf).max())


if __name__ == '__main__':
    main()
