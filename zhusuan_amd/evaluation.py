"""Annealed importance sampling on top of the native HMC -- the direct
*caller* of the hot path.  Mirrors reference
zhusuan/evaluation.py:57-172: tempered log-joint
(1 - T) * log_prior + T * log_joint with a sigmoid temperature schedule, HMC
transitions at each temperature, importance weights accumulated from
HMCInfo.orig_log_prob / log_prob, log-mean-exp lower bound.

Differences forced by having no TensorFlow: there is no session (the `sess`
argument of `run` is accepted and ignored) and the temperature is a host
scalar read by the tempered log-joint each time the sampler evaluates it (the
generic HMC plan re-evaluates the joint at every gradient evaluation, so a
changing temperature is honoured exactly like a fed placeholder)."""
import numpy as np
import torch

from .utils import merge_dicts

__all__ = ['AIS']


class AIS(object):
    """evaluation.py:57-110 (same constructor arguments)."""

    def __init__(self, meta_bn, proposal_meta_bn, hmc, observed, latent,
                 n_temperatures=1000, n_adapt=30, verbose=False):
        self._n_temperatures = n_temperatures
        self._n_adapt = n_adapt
        self._verbose = verbose
        if callable(meta_bn) and not hasattr(meta_bn, 'observe'):
            log_joint = meta_bn
        else:
            log_joint = lambda obs: meta_bn.observe(**obs).log_joint()
        self._latent_k, self._latent_v = zip(*latent.items())
        self._proposal = proposal_meta_bn
        log_prior = lambda obs: proposal_meta_bn.observe(**obs).log_joint()
        self.temperature = 0.0          # the tf.placeholder of :94-95

        def log_fn(observed_):
            t = float(self.temperature)
            # evaluation.py:98-100
            return log_prior(observed_) * (1 - t) + log_joint(observed_) * t

        self.log_fn = log_fn
        self._observed = dict(observed)
        self._latent = dict(latent)
        self.sample_op, self.hmc_info = hmc.sample(log_fn, observed, latent)
        self._hmc = hmc

    def _init_latent(self):
        """evaluation.py:96,109-110: draw the latents from the proposal."""
        prior_samples = self._proposal.observe().get(self._latent_k)
        for z, z_s in zip(self._latent_v, prior_samples):
            z.copy_(z_s.tensor if hasattr(z_s, 'tensor') else z_s)

    def _map_t(self, t):
        return 1. / (1. + np.exp(-4 * (2 * t / self._n_temperatures - 1)))

    def _get_schedule_t(self, t):
        return (self._map_t(t) - self._map_t(0)) / (
            self._map_t(self._n_temperatures) - self._map_t(0))

    def run(self, sess=None, feed_dict=None):
        """Run the AIS loop; returns the log marginal likelihood estimate
        (evaluation.py:119-165)."""
        adp_num_t = 2 if self._n_temperatures > 1 else 1
        adp_t = self._get_schedule_t(adp_num_t)
        self._init_latent()
        for i in range(self._n_adapt):
            self.temperature = adp_t
            self.sample_op.run(feed_dict=feed_dict)
            if self._verbose:
                print('Adapt iter {}, acc = {:.3f}'.format(
                    i, float(self.hmc_info.acceptance_rate.mean())))
        self._init_latent()
        self.temperature = 0.0
        with torch.no_grad():
            prior_density = self.log_fn(
                merge_dicts(self._observed, self._latent)).cpu().numpy()
        log_weights = -prior_density
        for num_t in range(self._n_temperatures):
            self.temperature = self._get_schedule_t(num_t + 1)
            self.sample_op.run(feed_dict=feed_dict)
            old_log_p = self.hmc_info.orig_log_prob.cpu().numpy()
            new_log_p = self.hmc_info.log_prob.cpu().numpy()
            if num_t + 1 < self._n_temperatures:
                log_weights = log_weights + old_log_p - new_log_p
            else:
                log_weights = log_weights + old_log_p
            if self._verbose:
                print('Finished step {}, Temperature = {:.4f}, acc = {:.3f}'
                      .format(num_t + 1, self.temperature, float(
                          self.hmc_info.acceptance_rate.mean())))
        return np.mean(self._get_lower_bound(log_weights))

    @staticmethod
    def _get_lower_bound(log_weights):
        """log-mean-exp over the leading (chain) axis, evaluation.py:167-172."""
        max_log_weights = np.max(log_weights, axis=0)
        offset_log_weights = np.mean(np.exp(log_weights - max_log_weights),
                                     axis=0)
        return np.log(offset_log_weights) + max_log_weights
