"""Annealed importance sampling on top of the native HMC -- the direct
*caller* of the hot path.  Same constructor and
`run()` as reference zhusuan/evaluation.py:57-172: the sampler targets
(1 - T) * log_prior + T * log_joint, T follows a sigmoid schedule from 0 to 1,
one HMC transition per temperature, and the importance weights are the
telescoping sums of HMCInfo.orig_log_prob / log_prob; the estimate is the
log-mean-exp over the chain axis.

When the proposal is the target's own model with the latent's prior as its
log-joint (lntm_mcem.py:128-134) and the target lowers to a native
dense-likelihood plan, the tempered target is `log prior + T * log lik` and
the sampler runs on that plan with the likelihood term scaled by T
(zshmc_model_kick_drift's lik_scale) -- no autograd graph per leapfrog trip.

MI355X-first differences: there is no session (`sess` is accepted and
ignored), the temperature is a host scalar that the tempered log-joint reads
every time the sampler evaluates it (exactly like a fed placeholder), and the
weights are accumulated **on the device** in stream order -- the
n_temperatures transitions are enqueued back to back with no host round trip
(the reference fetches two [chains] vectors per temperature), and one
synchronisation happens when the estimate is read."""
import math

import numpy as np
import torch

from . import _symbolic
from .utils import merge_dicts

__all__ = ['AIS']


def _as_log_joint(model):
    if callable(model) and not hasattr(model, 'observe'):
        return model
    return lambda values: model.observe(**values).log_joint()


def _proposal_is_prior_of(target, proposal, observed, latent):
    """True if `proposal` is the target's own model with the conditional
    log-density of the single latent as its log-joint (what lntm_mcem.py:128-134
    builds: `proposal = copy(model); proposal.log_joint = lambda bn:
    bn.cond_log_prob('eta')`) AND the target's log-joint is that prior plus
    one observed node's likelihood, in a form the native dense-likelihood
    plans take.  Checked structurally, on one evaluation of each."""
    from .framework.bn import StochasticTensor
    from .framework.meta_bn import MetaBayesianNet
    from .hmc import deferred, placeholder
    from .plans.recognise import _summands_of
    if not (isinstance(target, MetaBayesianNet) and
            isinstance(proposal, MetaBayesianNet)) or len(latent) != 1:
        return False
    def same(a, b):
        return len(a) == len(b) and all(x is y for x, y in zip(a, b))
    if getattr(proposal, '_builder', None) is not getattr(
            target, '_builder', 0) or \
            not same(proposal._call_args, target._call_args) or \
            sorted(proposal._call_kwargs) != sorted(target._call_kwargs) or \
            not same([proposal._call_kwargs[k]
                      for k in sorted(proposal._call_kwargs)],
                     [target._call_kwargs[k]
                      for k in sorted(target._call_kwargs)]):
        return False
    name, value = next(iter(latent.items()))
    obs = {k: (v.value if isinstance(v, (placeholder, deferred)) else v)
           for k, v in observed.items()}
    probe = value.detach().requires_grad_(True)
    try:
        sym_probe = _symbolic.wrap_latent(probe)   # literal dense spellings
        bn_p = proposal.observe(**merge_dicts(obs, {name: sym_probe}))
        lp = bn_p.log_joint()
        prior_node = bn_p.get(name)
        if not isinstance(prior_node, StochasticTensor) or \
                lp is not prior_node.__dict__.get('_cond_log_p'):
            return False
        bn_t = target.observe(**merge_dicts(obs, {name: sym_probe}))
        nodes = [n for n in bn_t.nodes.values()
                 if isinstance(n, StochasticTensor)]
        if target.log_joint is not None:
            nodes = _summands_of(bn_t.log_joint(), nodes)
        if nodes is None or len(nodes) != 2:
            return False
        return any(n.name == name for n in nodes) and any(
            n.name != name and n.is_observed() and
            getattr(n.dist, '_lazy', None) is not None for n in nodes)
    except Exception:                                    # noqa: BLE001
        return False


def sigmoid_schedule(n_temperatures):
    """T_0 = 0 .. T_n = 1 with T_k an affinely rescaled
    sigmoid(4 (2k/n - 1)) (evaluation.py:112-117)."""
    k = np.arange(n_temperatures + 1, dtype=np.float64)
    s = 1.0 / (1.0 + np.exp(-4.0 * (2.0 * k / n_temperatures - 1.0)))
    return (s - s[0]) / (s[-1] - s[0])


class AIS(object):
    def __init__(self, meta_bn, proposal_meta_bn, hmc, observed, latent,
                 n_temperatures=1000, n_adapt=30, verbose=False):
        self.n_temperatures = int(n_temperatures)
        self.n_adapt = int(n_adapt)
        self.verbose = bool(verbose)
        self.schedule = sigmoid_schedule(self.n_temperatures)
        self.temperature = 0.0          # the reference's tf.placeholder
        self._proposal = proposal_meta_bn
        self._names = tuple(latent.keys())
        self._state = tuple(latent.values())
        self._fixed = dict(observed)
        target, prior = _as_log_joint(meta_bn), _as_log_joint(proposal_meta_bn)

        def tempered(values):
            # the reference's temperature is a float32 placeholder: 1 - T is
            # formed in float32 (:101-103)
            t = np.float32(self.temperature)
            return prior(values) * float(np.float32(1) - t) + \
                target(values) * float(t)

        self.log_fn = tempered
        self.sample_op = self.hmc_info = None
        if _proposal_is_prior_of(meta_bn, proposal_meta_bn, observed, latent):
            # (1 - T) log prior + T (log prior + log lik) = log prior +
            # T log lik: the target's own native plan with the likelihood term
            # scaled by the temperature -- the fused MFMA likelihood instead
            # of an autograd graph per leapfrog trip (lntm_mcem.py:116-141)
            self.sample_op, self.hmc_info = hmc.sample(meta_bn, observed,
                                                       latent)
            plan = hmc._plan
            if hasattr(plan, 'lik_scale'):
                plan.lik_scale = lambda: float(np.float32(self.temperature))
            else:
                # the target did not lower to a dense-likelihood plan (shape,
                # alignment, native_plans=False): that plan could not anneal;
                # build the sampler again on the tempered callable
                hmc._plan = None
                self.sample_op = None
        if self.sample_op is None:
            self.sample_op, self.hmc_info = hmc.sample(tempered, observed,
                                                       latent)
        self._hmc = hmc

    # kept for callers of the reference's private helper
    def _get_schedule_t(self, k):
        return float(self.schedule[k])

    def _reset_state(self):
        """Latents <- one draw from the proposal (evaluation.py:96,109-110)."""
        draws = self._proposal.observe().get(self._names)
        for buf, node in zip(self._state, draws):
            buf.copy_(node.tensor if hasattr(node, 'tensor') else node)

    def _transition(self, k, feed_dict, tag):
        self.temperature = self.schedule[k]
        self.sample_op.run(feed_dict=feed_dict, sync=False)
        if self.verbose:
            print('{} {}, Temperature = {:.4f}, acc = {:.3f}'.format(
                tag, k, self.temperature,
                float(self.hmc_info.acceptance_rate.mean())))

    def run(self, sess=None, feed_dict=None):
        """The AIS loop (evaluation.py:119-165); returns the log marginal
        likelihood estimate averaged over the non-chain axes."""
        n = self.n_temperatures
        # step-size adaptation at a temperature close to the prior
        self._reset_state()
        for _ in range(self.n_adapt):
            self._transition(2 if n > 1 else 1, feed_dict, 'Adapt at step')
        # the annealing run proper, weights accumulated in stream order
        self._reset_state()
        self.temperature = 0.0
        with torch.no_grad():
            log_w = -self.log_fn(merge_dicts(
                self._fixed, dict(zip(self._names, self._state)))).clone()
        info = self.hmc_info
        plan = self._hmc._plan
        if hasattr(plan, 'lik_scale') and hasattr(plan, 'run_block') and \
                not self.verbose and log_w.is_contiguous():
            # native plan: the n annealing transitions and the weight updates
            # from ONE call into libzshmc.so (zshmc_hmc_model_run with the
            # temperature schedule as its lik_scale array) -- bit-identical
            # to the loop below
            self.sample_op.anneal(
                [float(np.float32(t)) for t in self.schedule[1:]],
                log_w, ends=True, feed_dict=feed_dict)
            self.temperature = self.schedule[n]
            n = 0
        for k in range(1, n + 1):
            self._transition(k, feed_dict, 'Finished step')
            log_w += info.orig_log_prob
            if k < n:
                log_w -= info.log_prob
        self._hmc.check_numerics()
        self.log_weights = log_w      # per chain, device (the reference keeps
        #                               them local to run())
        return float(self.lower_bound(log_w).mean())

    @staticmethod
    def lower_bound(log_weights):
        """log-mean-exp over the leading (chain) axis (evaluation.py:167-172);
        accepts a device tensor or an array."""
        if isinstance(log_weights, torch.Tensor):
            return torch.logsumexp(log_weights.double(), 0) - math.log(
                log_weights.shape[0])
        w = np.asarray(log_weights, np.float64)
        m = w.max(0)
        return np.log(np.mean(np.exp(w - m), 0)) + m

    _get_lower_bound = lower_bound
