#!/usr/bin/env python
"""The reference's examples/toy_examples/mixture_sgnht.py on zhusuan_amd:
1 000 chains of the stochastic-gradient Nose-Hoover thermostat (scalar
friction, first-order integrator) on a two-component 1-D Gaussian mixture.
Same sampler arguments; the log-joint is a plain torch function on the device
(its gradient comes from autograd), the thermostat update is the HIP kernel
behind zshmc_sgnht_update / zshmc_sgnht_scalar.

    python examples/mixture_sgnht.py [n_iters]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402


if __name__ == "__main__":
    zs.set_random_seed(1)
    torch.manual_seed(1)
    dev = torch.device('cuda', 0)

    # Define model parameters
    stdev = 0.5
    mu1 = -1
    mu2 = 3

    # Define sampler parameters
    n_chains = 1000
    n_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    burnin = n_iters * 2 // 3

    def log_joint(observed):
        x = observed['x']
        a1 = -0.5 * ((x - mu1) / stdev) ** 2
        a2 = -0.5 * ((x - mu2) / stdev) ** 2
        amax = torch.maximum(a1, a2)
        return amax + torch.log(torch.exp(a1 - amax) + torch.exp(a2 - amax))

    sgmcmc = zs.SGNHT(learning_rate=0.2, variance_extra=0.1, tune_rate=0.01,
                      second_order=False, use_vector_alpha=False)
    x = torch.rand(n_chains, device=dev) * 10 - 5
    sample_op, sgmcmc_info = sgmcmc.sample(log_joint, observed={},
                                           latent={'x': x})

    # Run the inference
    with zs.Session() as sess:
        samples = []
        print('Sampling...')
        for t in range(n_iters):
            if t % 500 == 0 or (t >= burnin and t % 100 == 0):
                _, info = sess.run([sample_op, sgmcmc_info])
                if t % 500 == 0:
                    print("mean_k: {}, alpha: {}".format(info.mean_k,
                                                         info.alpha))
                if t >= burnin and t % 100 == 0:
                    samples.append(info.q["x"])
            else:
                sample_op.run()           # enqueue only, no host round trip
        print('Finished.')
        samples = np.array(samples).reshape(-1)

    # Check the results
    total_stdev = np.sqrt(stdev**2 + 0.5 * (mu1**2 + mu2**2) -
                          (0.5 * (mu1 + mu2))**2)
    print('Expected mean = {}'.format(0.5 * (mu1 + mu2)))
    print('Sample mean = {}'.format(np.mean(samples)))
    print('Expected stdev = {}'.format(total_stdev))
    print('Sample stdev = {}'.format(np.std(samples)))
    rel = (np.std(samples) - total_stdev) / total_stdev
    print('Relative error of stdev = {}'.format(rel))
    hist, edges = np.histogram(samples, bins=40, range=(-4, 6), density=True)
    mid = 0.5 * (edges[1:] + edges[:-1])
    pdf = 0.5 * (np.exp(-0.5 * ((mid - mu1) / stdev) ** 2) +
                 np.exp(-0.5 * ((mid - mu2) / stdev) ** 2)) / \
        (stdev * np.sqrt(2 * np.pi))
    print('max |histogram - density| = {:.3f}'.format(np.abs(hist - pdf).max()))
