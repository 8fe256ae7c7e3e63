#!/usr/bin/env python
"""Golden HMC traces from the reference's OWN implementation, round 3: models
with SEVERAL latents feeding one dense likelihood --

    logits = tf.matmul(u, X1, transpose_b=True)
           + tf.matmul(v, X2, transpose_b=True) + tf.expand_dims(b, 1)

(two weight blocks of 7 and 6 features, a per-chain scalar intercept with a
group_ndims = 0 prior) -- the family the native plan's packed state samples;
and the topic model of lntm_mcem.py with K = 6 topics (rows padded to 8 on the
device).
Same machinery as oracle/make_golden_hmc.py (the reference's unmodified
zhusuan/hmc.py and model layer over oracle/tf_shim.py, the shared Philox
stream); writes tests/golden/hmc_reference_traces_r3.npz, which
tests/test_oracle_hmc_reference.py pins oracle/hmc_ref.py to.  Needs
/root/reference: run in the build container, commit the .npz.

    python -m oracle.make_golden_hmc_r3
"""
import os

import numpy as np

from oracle.hmc_case_data import blr_bias_data, lntm_ragged_data
from oracle.make_golden_hmc import ROOT, load_reference, lntm_model, run_case


def blr_bias_model(X1, X2):
    """u ~ N(0, 1), v ~ N(0, 0.5^2) (group_ndims 1), b ~ N(0, 2^2) per chain
    (group_ndims 0), y ~ Bernoulli(u X1^T + v X2^T + b) with group_ndims = 1,
    built with the reference's own bn.normal / bn.bernoulli (bn.py:556-590,
    628-654 -> univariate.py:43-184,334-406), tf.matmul and tf.expand_dims."""
    def make(tf, zs, n_chains):
        @zs.meta_bayesian_net()
        def blr():
            bn = zs.BayesianNet()
            u = bn.normal('u', tf.zeros([X1.shape[1]]), std=1.,
                          n_samples=n_chains, group_ndims=1)
            v = bn.normal('v', tf.zeros([X2.shape[1]]), std=0.5,
                          n_samples=n_chains, group_ndims=1)
            b = bn.normal('b', tf.zeros([]), std=2., n_samples=n_chains)
            logits = tf.matmul(u, tf.constant(X1), transpose_b=True) + \
                tf.matmul(v, tf.constant(X2), transpose_b=True) + \
                tf.expand_dims(b, 1)
            bn.bernoulli('y', logits, group_ndims=1)
            return bn
        return blr()
    return make


def cases():
    X1, X2, y, u0, v0, b0 = blr_bias_data()
    n = u0.shape[0]
    # step-size and mass adaptation fed per run; the search of the first
    # iteration and the re-initialisation at t = mass_collect_iters included
    return [dict(
        name='blr_bias', chain_shape=(n,),
        make_log_joint=blr_bias_model(X1, X2),
        make_observed=lambda tf: {'y': tf.constant(
            np.tile(y[None, :], (n, 1)))},
        latents={'u': u0, 'v': v0, 'b': b0},
        hmc_kwargs=dict(step_size=0.02, n_leapfrogs=5,
                        adapt_step_size='placeholder',
                        adapt_mass='placeholder',
                        target_acceptance_rate=0.8, mass_collect_iters=3),
        n_iters=12, flags=lambda i: (i < 10, i < 8), seed=19),
        _lntm_ragged()]


def _lntm_ragged():
    # the reference's own `lntm` (examples/topic_models/lntm_mcem.py:31-48,
    # imported) with K = 6 topics
    beta, x, eta_mean, eta_logstd, eta0 = lntm_ragged_data()
    n_c, n_d, n_k = eta0.shape
    return dict(
        name='lntm_k6', chain_shape=(n_c, n_d),
        make_log_joint=lntm_model(eta_mean, eta_logstd, n_d, n_k, x.shape[1]),
        make_observed=lambda tf: {
            'x': tf.constant(np.tile(x[None], (n_c, 1, 1))),
            'beta': tf.constant(beta)},
        latents={'eta': eta0},
        hmc_kwargs=dict(step_size=5e-3, n_leapfrogs=5,
                        adapt_step_size='placeholder',
                        adapt_mass='placeholder',
                        target_acceptance_rate=0.6, mass_collect_iters=3),
        n_iters=12, flags=lambda i: (i < 10, i < 8), seed=20)


def main():
    tf, zs = load_reference()
    res = {}
    for c in cases():
        res.update(run_case(tf, zs, **c))
        for k, v in c['latents'].items():
            res['%s/q0_%s' % (c['name'], k)] = v
    path = os.path.join(ROOT, 'tests', 'golden',
                        'hmc_reference_traces_r3.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, '(%d arrays)' % len(res))


if __name__ == '__main__':
    main()
