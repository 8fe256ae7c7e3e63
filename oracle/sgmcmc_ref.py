"""NumPy-float32 restatement of /root/reference/zhusuan/sgmcmc.py (SGLD,
PSGLD, SGHMC, SGNHT).  TEST INFRASTRUCTURE (see oracle/__init__.py).

PINNED (tests/test_oracle_sgmcmc_reference.py) against traces of the
reference's OWN zhusuan/sgmcmc.py, loaded unmodified from /root/reference and
run over the eager TensorFlow-API shim oracle/tf_shim.py on the stream below
(oracle/make_golden_sgmcmc.py -> tests/golden/sgmcmc_reference_traces.npz):
8 sampler configurations (SGLD, PSGLD, SGHMC and SGNHT first/second order,
vector/scalar friction, momentum resampling), two coupled latents.  The
reference's statistical tests (tests/test_mcmc.py:65-88, KDE bounds 0.023 /
0.016) are reproduced on the device path in tests/test_gpu_sgmcmc.py.  The Gaussian terms
(tf.random_normal, sgmcmc.py:200,250,311,318,326,446,455) come from the
shared Philox stream oracle/philox.py defines:
    counter (i//4 lo, i//4 hi, iteration, 3 | sub << 4 | latent_id << 8),
    sub 0 = step noise, sub 1 = momentum (re)sampling; the initial momentum
    (sgmcmc.py:310-314, :444-447) uses iteration 0xFFFFFFFF.
`t` is the value of SGMCMC.t BEFORE the run's assign_add (sgmcmc.py:76, :106):
the graph does not order the increment against the reads inside the update
ops, so "resample when t % n == 0" is evaluated on the pre-increment value
(the first run resamples).
"""
import numpy as np

from . import philox

F32 = np.float32
STREAM_SG = 3
SUB_NOISE, SUB_MOMENTUM = 0, 1
INIT_ITER = 0xFFFFFFFF


def _normal(seed, iteration, n, sub, latent_id):
    return philox.normal_flat(seed, iteration, n,
                              stream=STREAM_SG | (sub << 4) | (latent_id << 8))


class SGMCMC(object):
    """sgmcmc.py:24-166: `sample(grad, latent)` with `grad(list of q) ->
    list of d log p / dq` standing in for tf.gradients of the log joint."""

    def __init__(self, seed=0):
        self.t = 0
        self.seed = seed

    def sample(self, grad, latent):
        self._grad = grad
        self.qs = latent                      # list of float32 arrays, in place
        self._define_variables(self.qs)
        return self

    def step(self):
        info = self._update(self.qs, self._grad)
        self.t += 1                           # sgmcmc.py:106
        return info

    def _noise(self, k, q, std):
        return (_normal(self.seed, self.t, q.size, SUB_NOISE, k).reshape(q.shape)
                * F32(std)).astype(F32)

    def _momentum(self, k, q, std, iteration):
        return (_normal(self.seed, iteration, q.size, SUB_MOMENTUM, k)
                .reshape(q.shape) * F32(std)).astype(F32)


class SGLD(SGMCMC):
    """sgmcmc.py:169-204."""

    def __init__(self, learning_rate, seed=0):
        super(SGLD, self).__init__(seed)
        self.lr = F32(learning_rate)

    def _define_variables(self, qs):
        pass

    def _update(self, qs, grad):
        gs = grad(qs)
        for k, (q, g) in enumerate(zip(qs, gs)):
            q[...] = q + F32(0.5) * self.lr * g.astype(F32) + \
                self._noise(k, q, np.sqrt(self.lr))          # :200-201
        return {'q': qs}


class PSGLD(SGLD):
    """sgmcmc.py:207-253, RMSprop preconditioner (decay 0.9, epsilon 1e-3)."""

    def __init__(self, learning_rate, decay=0.9, epsilon=1e-3, seed=0):
        super(PSGLD, self).__init__(learning_rate, seed)
        self.decay, self.epsilon = F32(decay), F32(epsilon)

    def _define_variables(self, qs):
        self.vs = [np.zeros_like(q) for q in qs]             # :229-230

    def _update(self, qs, grad):
        gs = grad(qs)
        for k, (q, g, aux) in enumerate(zip(qs, gs, self.vs)):
            g = g.astype(F32)
            aux[...] = self.decay * aux + (F32(1) - self.decay) * g * g  # :234
            pre = F32(1) / (self.epsilon + np.sqrt(aux))               # :235
            z = _normal(self.seed, self.t, q.size, SUB_NOISE, k).reshape(q.shape)
            q[...] = q + F32(0.5) * self.lr * pre * g + \
                z * np.sqrt(self.lr * pre)                             # :249-250
        return {'q': qs}


class SGHMC(SGMCMC):
    """sgmcmc.py:256-363."""

    def __init__(self, learning_rate, friction=0.25, variance_estimate=0.,
                 n_iter_resample_v=20, second_order=True, seed=0):
        super(SGHMC, self).__init__(seed)
        self.lr = F32(learning_rate)
        self.alpha, self.beta = F32(friction), F32(variance_estimate)
        self.n_iter_resample_v = int(n_iter_resample_v or 0)
        self.second_order = bool(second_order)

    def _define_variables(self, qs):
        self.vs = [self._momentum(k, q, np.sqrt(self.lr), INIT_ITER)
                   for k, q in enumerate(qs)]                 # :310-314

    def _old_vs(self, qs):
        n = self.n_iter_resample_v
        if n != 0 and self.t % n == 0:                        # :319-326
            return [self._momentum(k, q, np.sqrt(self.lr), self.t)
                    for k, q in enumerate(qs)]
        return self.vs

    def _update(self, qs, grad):
        old_vs = self._old_vs(qs)
        std = np.sqrt(F32(2) * (self.alpha - self.beta) * self.lr)
        noises = [self._noise(k, q, std) for k, q in enumerate(qs)]   # :327-331
        mean_ks = []
        if not self.second_order:                             # :332-337
            gs = grad(qs)
            for k, (q, v, g, nz) in enumerate(zip(qs, old_vs, gs, noises)):
                nv = (F32(1) - self.alpha) * v + self.lr * g.astype(F32) + nz
                q[...] = q + nv
                self.vs[k] = nv.astype(F32)
                mean_ks.append(F32(np.mean(nv.astype(np.float64) ** 2)))
        else:                                                 # :338-347
            dh = np.exp(F32(-0.5) * self.alpha).astype(F32)
            for q, v in zip(qs, old_vs):
                q[...] = q + F32(0.5) * v                     # q1
            gs = grad(qs)
            for k, (q, v, g, nz) in enumerate(zip(qs, old_vs, gs, noises)):
                nv = dh * (dh * v + self.lr * g.astype(F32) + nz)
                q[...] = q + F32(0.5) * nv
                self.vs[k] = nv.astype(F32)
                mean_ks.append(F32(np.mean(nv.astype(np.float64) ** 2)))
        return {'q': qs, 'mean_k': mean_ks}


class SGNHT(SGHMC):
    """sgmcmc.py:366-497."""

    def __init__(self, learning_rate, variance_extra=0., tune_rate=1.,
                 n_iter_resample_v=None, second_order=True,
                 use_vector_alpha=True, seed=0):
        SGMCMC.__init__(self, seed)
        self.lr = F32(learning_rate)
        self.a, self.tune_rate = F32(variance_extra), F32(tune_rate)
        self.n_iter_resample_v = int(n_iter_resample_v or 0)
        self.second_order = bool(second_order)
        self.use_vector_alpha = bool(use_vector_alpha)

    def _define_variables(self, qs):
        SGHMC._define_variables(self, qs)
        if self.use_vector_alpha:                             # :448-450
            self.alphas = [np.full(q.shape, self.a, F32) for q in qs]
        else:
            self.alphas = [F32(self.a) for q in qs]

    def _mean(self, x):
        if self.use_vector_alpha:
            return x.astype(F32)
        return F32(np.mean(x.astype(np.float64)))

    def _update(self, qs, grad):
        old_vs = self._old_vs(qs)
        std = np.sqrt(F32(2) * self.a * self.lr)
        noises = [self._noise(k, q, std) for k, q in enumerate(qs)]   # :470-472
        mean_ks = []
        if not self.second_order:                             # :473-480
            gs = grad(qs)
            for k, (q, v, g, nz) in enumerate(zip(qs, old_vs, gs, noises)):
                al = self.alphas[k]
                nv = ((F32(1) - al) * v + self.lr * g.astype(F32) + nz).astype(F32)
                q[...] = q + nv
                self.vs[k] = nv
                mk = self._mean(nv * nv)
                mean_ks.append(mk)
                self.alphas[k] = (al + self.tune_rate * (mk - self.lr)).astype(F32)
        else:                                                 # :481-497
            a1s = []
            for k, (q, v) in enumerate(zip(qs, old_vs)):
                q[...] = q + F32(0.5) * v
                mk1 = self._mean(v * v)
                a1s.append((self.alphas[k] + F32(0.5) * self.tune_rate *
                            (mk1 - self.lr)).astype(F32))
            gs = grad(qs)
            for k, (q, v, g, nz) in enumerate(zip(qs, old_vs, gs, noises)):
                dh = np.exp(F32(-0.5) * a1s[k]).astype(F32)
                nv = (dh * (dh * v + self.lr * g.astype(F32) + nz)).astype(F32)
                q[...] = q + F32(0.5) * nv
                self.vs[k] = nv
                mk = self._mean(nv * nv)
                mean_ks.append(mk)
                self.alphas[k] = (a1s[k] + F32(0.5) * self.tune_rate *
                                  (mk - self.lr)).astype(F32)
        return {'q': qs, 'mean_k': mean_ks, 'alpha': list(self.alphas)}
