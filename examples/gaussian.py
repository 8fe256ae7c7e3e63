#!/usr/bin/env python
"""The reference's examples/toy_examples/gaussian.py (BASELINE config 1) on
zhusuan_amd: 1 000 chains of HMC on a diagonal Gaussian with step-size and
mass adaptation fed per iteration.  Same model definition, same sampler
arguments; `tf.*` becomes `torch.*` on the device, `sess.run([...], feed_dict)`
becomes `zs.Session().run([...], feed_dict)`.

    python examples/gaussian.py [n_x]
"""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402


@zs.meta_bayesian_net()
def gaussian(n_x, stdev, n_particles, device):
    bn = zs.BayesianNet()
    bn.normal('x', torch.zeros(n_x, device=device), std=stdev,
              n_samples=n_particles, group_ndims=1)
    return bn


if __name__ == "__main__":
    zs.set_random_seed(1)
    dev = torch.device('cuda', 0)

    # Define model parameters
    n_x = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    stdev = 1 / (np.arange(n_x, dtype=np.float32) + 1)

    # Define HMC parameters
    n_chains = 1000
    n_iters = 200
    burnin = n_iters // 2
    n_leapfrogs = 5

    # Build the sampler
    model = gaussian(n_x, torch.tensor(stdev, device=dev), n_chains, dev)
    adapt_step_size = zs.placeholder(bool, shape=[], name="adapt_step_size")
    adapt_mass = zs.placeholder(bool, shape=[], name="adapt_mass")
    hmc = zs.HMC(step_size=1e-3, n_leapfrogs=n_leapfrogs,
                 adapt_step_size=adapt_step_size, adapt_mass=adapt_mass,
                 target_acceptance_rate=0.9)
    x = torch.zeros(n_chains, n_x, device=dev)
    sample_op, hmc_info = hmc.sample(model, {}, {'x': x})
    print('plan:', hmc.plan_kind)

    # Run the inference
    sess = zs.Session()
    samples = []
    print('Sampling...')
    for i in range(n_iters):
        _, x_sample, acc, ss = sess.run(
            [sample_op, hmc_info.samples['x'], hmc_info.acceptance_rate,
             hmc_info.updated_step_size],
            feed_dict={adapt_step_size: i < burnin // 2,
                       adapt_mass: i < burnin // 2})
        if i % 20 == 0 or i < 12:
            print('Sample {}: Acceptance rate = {:.3f}, updated step size = '
                  '{:.5f}'.format(i, float(np.mean(acc)), float(ss)))
        if i >= burnin:
            samples.append(x_sample)
    print('Finished.')
    samples = np.vstack(samples)

    # Check the results
    print('Expected mean = {}'.format(np.zeros(n_x)))
    print('Sample mean = {}'.format(np.mean(samples, 0)))
    print('Expected stdev = {}'.format(stdev))
    print('Sample stdev = {}'.format(np.std(samples, 0)))
    rel = (np.std(samples, 0) - stdev) / stdev
    print('Relative error of stdev = {}'.format(rel))
    assert np.abs(np.mean(samples, 0)).max() < 0.05
    assert np.abs(rel).max() < 0.05
