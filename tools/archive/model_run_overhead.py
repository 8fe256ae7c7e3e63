#!/usr/bin/env python
"""Host cost of a transition of the native model plans: a Python loop of
`sample_op.run` against ONE `sample_op.run_many` / `anneal` call
(zshmc_hmc_model_run, csrc/hmc_model_run.hip), at the sizes the reference's
own loops run --
  * the E-step of examples/topic_models/lntm_mcem.py:157-182: a minibatch of
    100 documents, K = 100 topics, V = 12 419, 5 transitions of L = 20 per
    minibatch, step-size and mass adaptation on;
  * AIS.run (zhusuan/evaluation.py:119-165): 1 000 temperatures, 25 chains x
    300 held-out documents, L = 20 (lntm_mcem.py:208-219).
Prints wall microseconds per transition (device work included: the queue is
drained at both ends of each measurement) and the host-only enqueue time.
    python tools/archive/model_run_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zhusuan_amd as zs  # noqa: E402

dev = torch.device('cuda', 0)


def lntm(n_chains, n_docs, K, V, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
    mean = torch.zeros(n_docs, K, device=dev)
    logstd = torch.zeros(K, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', mean, logstd=logstd, n_samples=n_chains,
                        group_ndims=1)
        bn.unnormalized_multinomial(
            'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
            normalize_logits=False, dtype=torch.float32)
        return bn
    m = model()
    m.log_joint = lambda bn: bn.cond_log_prob('eta') + bn.cond_log_prob('x')
    return m, x


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6


def estep():
    model, x = lntm(1, 100, 100, 12419, 1)
    f = zs.placeholder(bool)
    out = {}
    for many in (False, True):
        hmc = zs.HMC(step_size=1e-3, n_leapfrogs=20, adapt_step_size=f,
                     adapt_mass=f, target_acceptance_rate=0.6, seed=1)
        eta = torch.zeros(1, 100, 100, device=dev)
        op, info = hmc.sample(model, {'x': x}, {'eta': eta})
        assert hmc.plan_kind == 'mixture_multinomial'
        for _ in range(15):                       # past the searches
            op.run(feed_dict={f: True}, sync=False)
        n_mb, per = 40, 5

        def loop():
            for _ in range(n_mb):
                if many:
                    op.run_many(per, feed_dict={f: True}, sync=False)
                else:
                    for _ in range(per):
                        op.run(feed_dict={f: True}, sync=False)
        loop()
        out[many] = timed(loop, n_mb * per)
    return out


def ais():
    import copy
    model, x = lntm(25, 300, 100, 12419, 2)
    proposal = copy.copy(model)
    proposal.log_joint = lambda bn: bn.cond_log_prob('eta')
    out = {}
    for block in (False, True):
        zs.set_random_seed(5)
        f = zs.placeholder(bool, default=False)
        hmc = zs.HMC(step_size=0.01, n_leapfrogs=20, adapt_step_size=f,
                     target_acceptance_rate=0.6, seed=3)
        eta = torch.zeros(25, 300, 100, device=dev)
        a = zs.AIS(model, proposal, hmc, {'x': x}, {'eta': eta},
                   n_temperatures=1000, n_adapt=30)
        assert hmc.plan_kind == 'mixture_multinomial'
        if not block:
            # the Python loop of evaluation.py:119-165 (what `verbose` keeps)
            a.verbose = True
            import builtins
            real = builtins.print
            builtins.print = lambda *args, **kw: None
        try:
            a.run(feed_dict={f: False})                       # warm
            out[block] = timed(lambda: a.run(feed_dict={f: False}), 1030)
        finally:
            if not block:
                builtins.print = real
    return out


if __name__ == '__main__':
    e = estep()
    print('E-step (100 docs x 100 topics x 12 419 words, L = 20, step size + '
          'mass adapting, 5 transitions per minibatch):')
    for many in (False, True):
        print('  %-44s %8.1f us per transition (host enqueue %7.1f us)' % (
            'sample_op.run_many(5) [zshmc_hmc_model_run]' if many
            else 'Python loop of sample_op.run', *e[many]))
    print('  wall ratio %.2fx, host ratio %.2fx' % (
        e[False][0] / e[True][0], e[False][1] / e[True][1]))
    a = ais()
    print('AIS (25 chains x 300 docs x 100 topics, 1 000 temperatures + 30 '
          'adaptation transitions, L = 20):')
    for block in (False, True):
        print('  %-44s %8.1f us per transition (host enqueue %7.1f us)' % (
            'sample_op.anneal [zshmc_hmc_model_run]' if block
            else 'Python loop (evaluation.py:119-165)', *a[block]))
    print('  wall ratio %.2fx, host ratio %.2fx' % (
        a[False][0] / a[True][0], a[False][1] / a[True][1]))
