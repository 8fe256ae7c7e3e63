"""NumPy restatement of /root/reference/zhusuan/evaluation.py:57-172 (AIS) on
top of oracle/hmc_ref.py.  TEST INFRASTRUCTURE (see oracle/__init__.py).
PINNED (tests/test_oracle_ais.py) to tests/golden/ais_reference.npz, a run of
the reference's own evaluation.py over oracle/tf_shim.py
(oracle/make_golden_ais.py).

The caller supplies log-prior / log-joint callables with analytic gradients
(tf.gradients in the reference) and a `draw_prior(k)` callable returning the
k-th proposal draw (proposal_meta_bn.observe().get(latent_k), :96)."""
import numpy as np


F32 = np.float32


class AIS(object):
    def __init__(self, log_prior, grad_prior, log_joint, grad_joint, hmc,
                 latent, draw_prior, n_temperatures=1000, n_adapt=30):
        self._n_temperatures = n_temperatures
        self._n_adapt = n_adapt
        self.temperature = F32(0.0)                 # tf.placeholder, :98
        self.latent = latent
        self._draw_prior = draw_prior
        self._n_draws = 0

        def log_fn(q):                              # :101-103
            t = self.temperature
            return (log_prior(q) * (F32(1) - t) + log_joint(q) * t).astype(F32)

        def grad_fn(q):                             # tf.gradients of log_fn
            t = self.temperature
            return [(gp * (F32(1) - t) + gj * t).astype(F32)
                    for gp, gj in zip(grad_prior(q), grad_joint(q))]
        self.log_fn = log_fn
        self.hmc = hmc.sample(log_fn, grad_fn, latent)     # :107-108

    def _map_t(self, t):                            # :112-113
        return 1. / (1. + np.exp(-4 * (2 * t / self._n_temperatures - 1)))

    def _get_schedule_t(self, t):                   # :115-117
        return (self._map_t(t) - self._map_t(0)) / (
            self._map_t(self._n_temperatures) - self._map_t(0))

    def _init_latent(self):                         # :109-110
        draws = self._draw_prior(self._n_draws)
        self._n_draws += 1
        for z, d in zip(self.latent, draws):
            z[...] = d

    def run(self):                                  # :119-165
        adp_num_t = 2 if self._n_temperatures > 1 else 1
        adp_t = self._get_schedule_t(adp_num_t)
        self._init_latent()
        self.acceptance = []
        for _ in range(self._n_adapt):
            self.temperature = F32(adp_t)
            info = self.hmc.step()
            self.acceptance.append(info.acceptance_rate.copy())
        self._init_latent()
        self.temperature = F32(0)
        log_weights = -self.log_fn(self.latent)
        for num_t in range(self._n_temperatures):
            self.temperature = F32(self._get_schedule_t(num_t + 1))
            info = self.hmc.step()
            self.acceptance.append(info.acceptance_rate.copy())
            if num_t + 1 < self._n_temperatures:
                log_weights = log_weights + info.orig_log_prob - info.log_prob
            else:
                log_weights = log_weights + info.orig_log_prob
        self.log_weights = log_weights
        return np.mean(self._get_lower_bound(log_weights))

    @staticmethod
    def _get_lower_bound(log_weights):              # :167-172
        max_log_weights = np.max(log_weights, axis=0)
        offset_log_weights = np.mean(np.exp(log_weights - max_log_weights),
                                     axis=0)
        return np.log(offset_log_weights) + max_log_weights
