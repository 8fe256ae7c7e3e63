"""Annealed importance sampling on top of the native HMC -- the direct
*caller* of the hot path.  Same constructor and
`run()` as reference zhusuan/evaluation.py:57-172: the sampler targets
(1 - T) * log_prior + T * log_joint, T follows a sigmoid schedule from 0 to 1,
one HMC transition per temperature, and the importance weights are the
telescoping sums of HMCInfo.orig_log_prob / log_prob; the estimate is the
log-mean-exp over the chain axis.

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

from .utils import merge_dicts

__all__ = ['AIS']


def _as_log_joint(model):
    if callable(model) and not hasattr(model, 'observe'):
        return model
    return lambda values: model.observe(**values).log_joint()


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
        self.sample_op, self.hmc_info = hmc.sample(tempered, observed, latent)
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
