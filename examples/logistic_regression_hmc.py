#!/usr/bin/env python
"""BASELINE config 3: Bayesian logistic regression by many-chain HMC.
A synthetic N x D design matrix, w ~ N(0, I), y ~ Bernoulli(sigmoid(X w)).
The likelihood is written as in the reference, `logits = w @ X^T`
(tf.matmul(w, X, transpose_b=True)): the sampler hands the model function a
symbolic latent (zhusuan_amd/_symbolic.py), the matmul stays symbolic, and
`bn.bernoulli` lowers it to the fused fp32-MFMA kernel -- the Bernoulli
log-likelihood of all chains and its gradient in one pass over X; the
[n_chains, N] logits (131 GB at the default size) never exist in memory.
(`zs.linear_logits(w, X)` is the explicit spelling of the same thing.)
`--bias` adds a per-chain intercept b ~ N(0, 2^2), `logits = w @ X^T +
b[:, None]`: two latents, still one fused likelihood (the sum lowers to one
lazy operand; the sampler packs w and b side by side), and `--d` may be
anything up to 1 024 -- not only a multiple of 4 below 256.

    python examples/logistic_regression_hmc.py [--n 1000000] [--d 256]
        [--chains 32768] [--iters 100] [--bias]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=100000)
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--chains', type=int, default=4096)
    ap.add_argument('--iters', type=int, default=100)
    ap.add_argument('--leapfrogs', type=int, default=10)
    ap.add_argument('--bias', action='store_true',
                    help='add a per-chain intercept latent')
    args = ap.parse_args()
    zs.set_random_seed(7)
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)

    N, D, C = args.n, args.d, args.chains
    X = torch.randn(N, D, device=dev, generator=g) / D ** 0.5
    w_true = torch.randn(D, device=dev, generator=g) * 2.0
    b_true = 0.5 if args.bias else 0.0
    y = (torch.rand(N, device=dev, generator=g) <
         torch.sigmoid(X @ w_true + b_true)).to(torch.float32)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D, device=dev), std=1.,
                      n_samples=C, group_ndims=1)
        logits = w.tensor @ X.t()
        if args.bias:
            b = bn.normal('b', torch.zeros((), device=dev), std=2.,
                          n_samples=C)
            logits = logits + b.tensor[:, None]
        bn.bernoulli('y', logits, group_ndims=1, dtype=torch.float32)
        return bn

    model = blr()
    adapt = zs.placeholder(bool, shape=[], name='adapt')
    hmc = zs.HMC(step_size=1e-3, n_leapfrogs=args.leapfrogs,
                 adapt_step_size=adapt, adapt_mass=adapt,
                 target_acceptance_rate=0.8)
    w = torch.zeros(C, D, device=dev)
    latent = {'w': w}
    if args.bias:
        latent['b'] = torch.zeros(C, device=dev)
    sample_op, info = hmc.sample(model, {'y': y}, latent)
    print('plan:', hmc.plan_kind)
    burnin = args.iters // 2
    draws = []
    torch.cuda.synchronize()
    t_start = time.time()
    for i in range(args.iters):
        sample_op.run(feed_dict={adapt: i < burnin})
        if i % 10 == 0 or i == args.iters - 1:
            print('iter {:3d}: acc = {:.3f}  step size = {:.5f}  '
                  'log p = {:.1f}'.format(
                      i, info.acceptance_rate.mean().item(),
                      float(info.updated_step_size),
                      info.log_prob.mean().item()))
        if i >= burnin:
            draws.append(torch.cat([w.mean(0), latent['b'].mean(0, True)])
                         if args.bias else w.mean(0))
    torch.cuda.synchronize()
    dt = time.time() - t_start
    evals = args.iters * (args.leapfrogs + 1)
    print('{:.2f} s; {:.3g} chain-leapfrog-steps/s; likelihood+gradient '
          '{:.1f} TFLOP/s sustained over the whole loop'.format(
              dt, C * args.leapfrogs * args.iters / dt,
              4.0 * C * N * D * evals / dt / 1e12))
    post_mean = torch.stack(draws).mean(0)
    # Laplace check: the posterior mean should sit near the MAP estimate
    # (with --bias the last entry is the intercept, prior N(0, 2^2))
    wm = torch.zeros(D + int(args.bias), device=dev, requires_grad=True)
    opt = torch.optim.LBFGS([wm], max_iter=200, line_search_fn='strong_wolfe')

    def closure():
        opt.zero_grad()
        z = X @ wm[:D] + (wm[D] if args.bias else 0.0)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(
            z, y, reduction='sum') + 0.5 * (wm[:D] ** 2).sum()
        if args.bias:
            loss = loss + 0.5 * (wm[D] / 2.0) ** 2
        loss.backward()
        return loss
    opt.step(closure)
    err = (post_mean - wm.detach()).norm() / wm.detach().norm()
    print('|posterior mean - MAP| / |MAP| = {:.3f}'.format(err.item()))
    if args.iters >= 60:          # a converged run: posterior mean near the MAP
        assert err.item() < 0.2
