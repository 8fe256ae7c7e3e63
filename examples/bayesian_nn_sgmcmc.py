#!/usr/bin/env python
"""Bayesian neural-network regression by stochastic-gradient MCMC (the
workload of the reference's BNN SGMCMC example): a one-hidden-layer network
whose weight matrices are the latents, `n_particles` parallel chains, a
mini-batch log joint rescaled to the training-set size, SGHMC (default),
SGLD or SGNHT as the sampler, and an M step that re-estimates the prior
log-stds of the weights from the particles after every epoch.

What it exercises on the device: several latents of different shapes in one
sampler, `group_ndims = 2` Normal priors, a deterministic node, a user
log-joint assembled from `cond_log_prob`, placeholders fed with a new
mini-batch on every run, hyper-parameters updated in place between runs, and
`sgmcmc_info.mean_k`.  The element-wise sampler updates are the HIP kernels
of csrc/sgmcmc.hip; the network itself is ordinary torch code (its gradient
is the sampler's input, not part of the sampler).

No data set is reachable offline: the regression problem is synthetic (a
random two-layer teacher network with observation noise), shaped like the UCI
protein set the reference example loads (9 features); --small shrinks it.

    python examples/bayesian_nn_sgmcmc.py [--small] [--sampler sghmc|sgld|sgnht]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402

Y_LOGSTD = -0.95          # observation noise of the model (log std)


def teacher_data(n, x_dim, rng, noise=0.3):
    w1 = rng.normal(size=(x_dim, 16)) / math.sqrt(x_dim)
    w2 = rng.normal(size=16) / 4.0
    x = rng.normal(size=(n, x_dim)).astype(np.float32)
    y = np.tanh(x @ w1) @ w2 + noise * rng.normal(size=n)
    return x, y.astype(np.float32)


def standardize(train, test):
    mean, std = train.mean(0), train.std(0)
    return (train - mean) / std, (test - mean) / std, mean, std


def make_model(x_in, layer_sizes, prior_logstds, n_particles):
    """The network as a MetaBayesianNet: one Normal node per weight matrix
    (bias folded in as an extra input column), activations scaled by
    1/sqrt(fan-in), a deterministic node for the prediction and a Normal
    observation node."""
    pairs = list(zip(layer_sizes[:-1], layer_sizes[1:]))

    @zs.meta_bayesian_net(scope='bnn', reuse_variables=True)
    def build():
        bn = zs.BayesianNet()
        x = x_in.value
        h = x.unsqueeze(0).expand(n_particles, -1, -1)
        for i, (n_in, n_out) in enumerate(pairs):
            w = bn.normal('w%d' % i,
                          torch.zeros(n_out, n_in + 1, device=x.device),
                          logstd=prior_logstds[i], group_ndims=2,
                          n_samples=n_particles)
            h = torch.cat([h, torch.ones_like(h[..., :1])], -1)
            h = torch.einsum('imk,ijk->ijm', w.tensor, h) / math.sqrt(n_in + 1)
            if i < len(pairs) - 1:
                h = torch.relu(h)
        y_mean = bn.deterministic('y_mean', h.squeeze(2))
        bn.normal('y', y_mean, logstd=Y_LOGSTD)
        return bn

    return build()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--sampler', default='sghmc',
                    choices=['sghmc', 'sgld', 'sgnht'])
    args = ap.parse_args()
    zs.set_random_seed(1237)
    torch.manual_seed(1237)
    rng = np.random.RandomState(2345)
    device = torch.device('cuda', 0)

    n_train, n_test, x_dim = (4000, 1000, 9) if args.small else (40000, 5000, 9)
    epochs = args.epochs or (8 if args.small else 30)
    x_all, y_all = teacher_data(n_train + n_test, x_dim, rng)
    x_train, x_test, _, _ = standardize(x_all[:n_train], x_all[n_train:])
    y_train, y_test, _, y_std = standardize(y_all[:n_train], y_all[n_train:])

    n_particles, batch = 20, 100
    layer_sizes = [x_dim, 50, 1]
    names = ['w%d' % i for i in range(len(layer_sizes) - 1)]
    x = zs.placeholder(torch.float32, name='x')
    y = zs.placeholder(torch.float32, name='y')
    x.feed(x_train[:batch], device)
    y.feed(y_train[:batch], device)
    particles, prior_logstds = [], []
    for n_in, n_out in zip(layer_sizes[:-1], layer_sizes[1:]):
        particles.append(torch.empty(n_particles, n_out, n_in + 1,
                                     device=device).uniform_(-2.0, 2.0))
        prior_logstds.append(torch.zeros(n_out, n_in + 1, device=device))

    model = make_model(x, layer_sizes, prior_logstds, n_particles)

    def log_joint(bn):
        log_pw = bn.cond_log_prob(names)
        log_py = bn.cond_log_prob('y')             # [particles, batch]
        return sum(log_pw) + log_py.mean(1) * n_train

    model.log_joint = log_joint

    if args.sampler == 'sgld':
        sampler = zs.SGLD(learning_rate=4e-6)
    elif args.sampler == 'sgnht':
        sampler = zs.SGNHT(learning_rate=1e-5, variance_extra=0.,
                           tune_rate=50., second_order=True)
    else:
        sampler = zs.SGHMC(learning_rate=2e-6, friction=0.2,
                           n_iter_resample_v=1000, second_order=True)
    latent = dict(zip(names, particles))
    step, info = sampler.sample(model, observed={'y': y}, latent=latent)

    x_test_d = torch.from_numpy(x_test).to(device)
    y_test_d = torch.from_numpy(y_test).to(device)
    n_batches = (n_train - 1) // batch + 1
    rmse = []
    for epoch in range(1, epochs + 1):
        order = rng.permutation(n_train)
        x_train, y_train = x_train[order], y_train[order]
        x_dev = torch.from_numpy(x_train).to(device)
        y_dev = torch.from_numpy(y_train).to(device)
        for b in range(n_batches):
            rows = slice(b * batch, (b + 1) * batch)
            step.run(feed_dict={x: x_dev[rows], y: y_dev[rows]})
        # M step: prior log-stds from the particles, written in place
        for w, logstd in zip(particles, prior_logstds):
            logstd.copy_(0.5 * torch.log((w * w).mean(0)))
        # posterior-mean prediction over the particles
        x.feed(x_test_d, device)
        with torch.no_grad():
            pred = model.observe(**latent)['y_mean'].mean(0)
        rmse.append(float(torch.sqrt(((pred - y_test_d) ** 2).mean())) *
                    float(y_std))
        kin = ''
        if hasattr(info, 'mean_k'):
            kin = ', mean_k = ' + ' '.join(
                '%.3g' % float(info.mean_k[k].mean()) for k in names)
        print('>> Epoch {} Test rmse = {:.4f}{}, prior logstd = {}'.format(
            epoch, rmse[-1], kin, ' '.join(
                '%.3f' % float(0.5 * torch.log((w * w).mean()))
                for w in particles)))
    print('noise floor (teacher) = {:.4f}'.format(0.3))


if __name__ == '__main__':
    main()
