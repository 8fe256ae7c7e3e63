#!/usr/bin/env python
"""Golden AIS run from the reference's OWN implementation.

Runs /root/reference/zhusuan/evaluation.py:AIS -- unmodified, with the
reference's own hmc.py and model layer under it, all over oracle/tf_shim.py
-- on a conjugate Gaussian model and records the per-chain log importance
weights, the estimate and the acceptance trace into
tests/golden/ais_reference.npz.  tests/test_oracle_ais.py pins
oracle/ais_ref.py to it, tests/test_gpu_lntm_ais.py the device path.

A TensorFlow-1 `sess.run(fetches, feed_dict)` re-executes a graph; in eager
execution the harness session below re-executes what AIS.__init__ built, from
AIS's own closures: fetching `sample_op` calls hmc.sample(ais.log_fn, ...)
again (variables replayed, as in make_golden_hmc.py), fetching `init_latent`
re-draws the proposal and assigns it, fetching `log_fn_val` re-evaluates
ais.log_fn.  AIS.__init__, AIS.run, the temperature schedule, the weight
accumulation and the log-mean-exp bound are the reference's code.

Random numbers: momenta / MH uniforms from the sampler's Philox mapping (seed
= HMC seed); proposal draws from the stand-alone sampling-op mapping
(philox.normal_flat(global seed, op offset)), offsets 0 and 1 for the two
executions of `init_latent` in AIS.run -- what zhusuan_amd draws after
set_random_seed(global seed).

    python -m oracle.make_golden_ais
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import philox, tf_shim  # noqa: E402
from oracle.make_golden_hmc import Stream, load_reference  # noqa: E402

# the case (mirrored by tests/helpers_ais_case.py)
N_CHAINS, D = 64, 6
N_TEMPERATURES, N_ADAPT = 40, 8
HMC_SEED, GLOBAL_SEED = 31, 5
W = np.linspace(0.5, 1.5, D).astype(np.float32)
X_STD = np.float32(0.7)
X_OBS = (np.random.RandomState(3).normal(size=D) * 1.2).astype(np.float32)
HMC_KW = dict(step_size=0.05, n_leapfrogs=5, adapt_step_size=True,
              target_acceptance_rate=0.7)


class HarnessSession(object):
    def __init__(self, tf, ais, hmc, observed, latent, proposal, stream,
                 prior_offset, global_seed=GLOBAL_SEED):
        self.global_seed = global_seed
        self.tf, self.ais, self.hmc = tf, ais, hmc
        self.observed, self.latent, self.proposal = observed, latent, proposal
        self.stream = stream
        self.prior_offset = prior_offset        # [next op offset]
        self.it = 0
        self.mark = None
        self.acc_trace = []
        self._fields = {id(getattr(ais.hmc_info, f)): f for f in (
            'acceptance_rate', 'orig_log_prob', 'log_prob')}

    def _prior_normal(self, shape):
        n = int(np.prod(shape))
        z = philox.normal_flat(self.global_seed, self.prior_offset[0], n)
        self.prior_offset[0] += 1
        return z.reshape(shape)

    def run(self, fetches, feed_dict=None):
        for k, v in (feed_dict or {}).items():
            k.feed(v)
        ais = self.ais
        if fetches is ais.init_latent:
            tf_shim.set_random_source(self._prior_normal, None)
            names = list(self.latent.keys())
            draws = self.proposal.observe().get(names)
            return [self.tf.assign(self.latent[n], d)
                    for n, d in zip(names, draws)]
        if fetches is ais.log_fn_val:
            from zhusuan.utils import merge_dicts
            return ais.log_fn(merge_dicts(self.observed, self.latent)
                              ).detach().numpy()
        assert fetches[0] is ais.sample_op
        self.it += 1
        self.stream.begin(self.it)
        tf_shim.set_random_source(self.stream.normal, self.stream.uniform)
        tf_shim.begin_run(self.mark)
        _, info = self.hmc.sample(ais.log_fn, self.observed, self.latent)
        tf_shim.end_replay()
        self.acc_trace.append(info.acceptance_rate.detach().numpy().copy())
        return [None] + [getattr(info, self._fields[id(f)]).detach().numpy()
                         for f in fetches[1:]]


def build(tf, zs):
    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        z = bn.normal('z', tf.zeros([D]), std=1., n_samples=N_CHAINS,
                      group_ndims=1)
        bn.normal('x', z * tf.constant(W), std=X_STD, group_ndims=1)
        return bn

    @zs.meta_bayesian_net()
    def proposal():
        bn = zs.BayesianNet()
        bn.normal('z', tf.zeros([D]), std=1., n_samples=N_CHAINS,
                  group_ndims=1)
        return bn
    return model(), proposal()


def true_log_marginal():
    var = W.astype(np.float64) ** 2 + float(X_STD) ** 2
    return float(np.sum(-0.5 * np.log(2 * np.pi * var)
                        - 0.5 * X_OBS.astype(np.float64) ** 2 / var))


def run_case(tf, zs, build, latent0, make_observed, chain_shape, hmc_kw,
             n_temperatures, n_adapt, hmc_seed, global_seed):
    """One run of the reference's AIS; returns (estimate, arrays)."""
    tf_shim._VARS[:] = []
    tf_shim.end_replay()
    model, proposal = build(tf, zs)
    latent = {k: tf.Variable(v.copy(), name=k) for k, v in latent0.items()}
    observed = make_observed(tf)
    hmc = zs.hmc.HMC(**hmc_kw)
    stream = Stream(hmc_seed, chain_shape)
    stream.begin(0)
    offset = [0]
    # AIS.__init__ builds the graph: its proposal draw and its one
    # hmc.sample() are graph construction in TensorFlow and consume nothing;
    # here they execute eagerly on throw-away numbers
    throwaway = np.random.RandomState(0)
    tf_shim.set_random_source(
        lambda shape: throwaway.normal(size=shape).astype(np.float32),
        lambda shape: throwaway.uniform(size=shape).astype(np.float32))
    mark = tf_shim.variable_mark()
    state_before = [v.numpy() for v in tf_shim._VARS]
    tf_shim.Placeholder.unfed_default = 0.0
    ais = zs.evaluation.AIS(model, proposal, hmc, observed, latent,
                            n_temperatures=n_temperatures, n_adapt=n_adapt)
    tf_shim.Placeholder.unfed_default = None
    tf_shim.end_replay()
    # undo what the eager "graph construction" executed: latents, sampler
    # variables (t, step size, tuner) back to their initial values
    for v, init in zip(tf_shim._VARS, state_before):
        v.assign(init)
    for k, v in latent0.items():
        latent[k].assign(v)
    _reset_sampler_variables(hmc, hmc_kw)
    sess = HarnessSession(tf, ais, hmc, observed, latent, proposal, stream,
                          offset, global_seed)
    sess.mark = mark
    # capture the per-chain weights: AIS.run only returns their bound
    captured = {}
    orig_bound = ais._get_lower_bound

    def capture(log_weights):
        captured['log_weights'] = np.array(log_weights, np.float32)
        return orig_bound(log_weights)
    ais._get_lower_bound = capture
    estimate = ais.run(sess, feed_dict={})
    out = {
        'estimate': np.float64(estimate),
        'log_weights': captured['log_weights'],
        'acceptance_rate': np.stack(sess.acc_trace),
        'final_step_size': np.float32(
            tf_shim._t(hmc.step_size).detach().numpy()),
    }
    for k in latent0:
        out[k + '_final'] = latent[k].numpy()
    return estimate, out


# ---- second case: the evaluation block of lntm_mcem.py (:116-141) -----------
LNTM_N_TEMPERATURES, LNTM_N_ADAPT = 10, 4
LNTM_HMC_SEED, LNTM_GLOBAL_SEED = 33, 7
LNTM_HMC_KW = dict(step_size=0.01, n_leapfrogs=5, adapt_step_size=True,
                   target_acceptance_rate=0.6)


def build_lntm(tf, zs):
    """The reference's OWN `lntm` model function (imported from the
    unmodified examples/topic_models/lntm_mcem.py) with the script's E-step
    objective (:97-102) as the target and its copy with the prior of eta as
    the proposal (:128-134)."""
    from copy import copy
    from oracle.hmc_case_data import lntm_data
    from oracle.make_golden_hmc import load_reference_example
    beta, x, eta_mean, eta_logstd, eta0 = lntm_data()
    n_chains, n_docs, n_topics = eta0.shape
    example = load_reference_example('topic_models/lntm_mcem.py')
    model = example.lntm(n_chains, n_docs, n_topics, x.shape[1],
                         tf.constant(eta_mean), tf.constant(eta_logstd))
    model.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
    proposal = copy(model)
    proposal.log_joint = lambda bn: bn.cond_log_prob('eta')
    return model, proposal


def main():
    tf, zs = load_reference()
    estimate, out = run_case(
        tf, zs, build, {'z': np.zeros((N_CHAINS, D), np.float32)},
        lambda tf: {'x': tf.constant(X_OBS)}, (N_CHAINS,), HMC_KW,
        N_TEMPERATURES, N_ADAPT, HMC_SEED, GLOBAL_SEED)
    out.update({'true_log_marginal': np.float64(true_log_marginal()),
                'x_obs': X_OBS, 'w': W})
    path = os.path.join(ROOT, 'tests', 'golden', 'ais_reference.npz')
    np.savez_compressed(path, **out)
    print('AIS estimate %.5f  (exact log marginal %.5f), mean acc %.3f, '
          'final eps %.4f' % (estimate, out['true_log_marginal'],
                              out['acceptance_rate'][N_ADAPT:].mean(),
                              out['final_step_size']))
    print('wrote', path)

    from oracle.hmc_case_data import lntm_data
    beta, x, _, _, eta0 = lntm_data()
    # (the counts are given with the full batch shape: distributions/
    # utils.py:43-44 broadcasts with an in-place multiply, see
    # make_golden_hmc.py case F)
    estimate, out = run_case(
        tf, zs, build_lntm, {'eta': np.zeros_like(eta0)},
        lambda tf: {'x': tf.constant(np.tile(x[None], (eta0.shape[0], 1, 1))),
                    'beta': tf.constant(beta)},
        eta0.shape[:2], LNTM_HMC_KW, LNTM_N_TEMPERATURES, LNTM_N_ADAPT,
        LNTM_HMC_SEED, LNTM_GLOBAL_SEED)
    path = os.path.join(ROOT, 'tests', 'golden', 'ais_lntm_reference.npz')
    np.savez_compressed(path, **out)
    print('lntm AIS estimate %.5f per document, mean acc %.3f, final eps %.4f'
          % (estimate, out['acceptance_rate'][LNTM_N_ADAPT:].mean(),
             out['final_step_size']))
    print('wrote', path)


def _reset_sampler_variables(hmc, kw):
    """hmc.py:258-264 / :79-87 initial values."""
    hmc.step_size.assign(np.float32(kw['step_size']))
    hmc.t.assign(np.float32(0.0))
    tuner = hmc.step_size_tuner
    tuner.step.assign(np.float32(0.0))
    tuner.log_epsilon_bar.assign(np.float32(0.0))
    tuner.h_bar.assign(np.float32(0.0))


if __name__ == '__main__':
    main()
