"""Stochastic-gradient MCMC on the MI355X: SGLD, PSGLD, SGHMC, SGNHT with the
surface of reference zhusuan/sgmcmc.py (same class names, constructor
arguments and defaults, `sample(meta_bn, observed, latent) -> (sample_op,
sgmcmc_info)`, `sgmcmc_info.q / .mean_k / .alpha` dictionaries keyed by latent
name).  The element-wise updates and their Gaussian terms are HIP kernels
(csrc/sgmcmc.hip) behind the C-ABI; the gradient of the user's (mini-batch)
log joint comes from torch autograd, the role tf.gradients plays in
sgmcmc.py:95-99.  No CPU fallback.

Differences forced by the absence of TensorFlow: latents are float32 device
tensors updated in place; `sample_op.run(feed_dict)` executes one iteration;
hyper-parameters accept Python numbers or `zhusuan_amd.placeholder` objects
(fed per run, e.g. a decaying learning rate, sgmcmc.py:186-188).
`t` counts completed runs; momentum is resampled when t % n_iter_resample_v
== 0 evaluated BEFORE the run's increment (the reference graph leaves the
order of `t.assign_add(1)` and the reads of `t` unspecified, sgmcmc.py:106).
"""
from collections import namedtuple
import math

import torch

from . import _capi, _writes
from .framework.meta_bn import MetaBayesianNet
from .hmc import bind_feed, deferred, placeholder
from .utils import merge_dicts, next_sampler_seed

__all__ = ['SGMCMC', 'SGLD', 'PSGLD', 'SGHMC', 'SGNHT']

_INIT_ITER = 0xFFFFFFFF


def _value(x, feed_dict, what):
    """Python number or placeholder -> float for this run."""
    if isinstance(x, placeholder):
        if feed_dict is None or x not in feed_dict:
            if x.default is not None:
                return float(x.default)
            raise ValueError('%s is a placeholder and was not fed' % what)
        return float(feed_dict[x])
    return float(x)


class _SampleOp(object):
    def __init__(self, sampler):
        self._sampler = sampler

    def run(self, feed_dict=None, sync=False):
        self._sampler._run(feed_dict)
        if sync:
            torch.cuda.current_stream().synchronize()

    __call__ = run


class SGMCMC(object):
    """Base class (sgmcmc.py:24-166)."""

    def __init__(self, seed=None):
        self.t = 0                                            # sgmcmc.py:76
        self.seed = next_sampler_seed() if seed is None else \
            int(seed) & 0xFFFFFFFFFFFFFFFF
        self._built = False

    # -- sgmcmc.py:78-100 ---------------------------------------------------
    def _make_grad_func(self, meta_bn, observed, latent):
        if callable(meta_bn) and not isinstance(meta_bn, MetaBayesianNet):
            self._log_joint = meta_bn
        else:
            self._log_joint = lambda obs: meta_bn.observe(**obs).log_joint()
        self._observed = dict(observed)
        latent_k, latent_v = [list(i) for i in zip(*latent.items())]
        for k, v in zip(latent_k, latent_v):
            if not isinstance(v, torch.Tensor):
                raise TypeError("latent['{}'] is not a torch Tensor (the "
                                "device buffer that replaces a tensorflow "
                                "Variable).".format(k))
            if v.dtype != torch.float32 or not v.is_cuda or \
                    not v.is_contiguous() or v.requires_grad:
                raise ValueError("latent['{}'] must be a contiguous float32 "
                                 "device tensor without grad.".format(k))
        self._latent_k = latent_k
        self._var_list = latent_v

        def grad_func(var_list):
            leaves = [q.detach().requires_grad_(True) for q in var_list]
            # placeholders / deferred expressions in `observed` resolve to the
            # values bound by this run's feed_dict (the mini-batch idiom
            # sess.run(sample_op, feed_dict={x: xb, y: yb}))
            observed_now = {
                k: (v.value if isinstance(v, (placeholder, deferred)) else v)
                for k, v in self._observed.items()}
            joint_obs = merge_dicts(dict(zip(latent_k, leaves)), observed_now)
            lp = self._log_joint(joint_obs)
            grads = torch.autograd.grad(lp.sum(), leaves, allow_unused=True)
            return [torch.zeros_like(q) if g is None else
                    g.to(torch.float32).contiguous()
                    for g, q in zip(grads, leaves)]
        return grad_func

    def sample(self, meta_bn, observed, latent):
        """sgmcmc.py:115-160: returns `(sample_op, sgmcmc_info)`."""
        if self._built:
            raise RuntimeError('sample may be invoked once per sampler')
        self._grad_func = self._make_grad_func(meta_bn, observed, latent)
        self._define_variables(self._var_list)
        self._built = True
        infos = self._info_dicts()
        names = list(infos.keys())
        info_t = namedtuple('SGMCMCInfo', names)
        self.sgmcmc_info = info_t(**infos)
        return _SampleOp(self), self.sgmcmc_info

    def _info_dicts(self):
        return {'q': dict(zip(self._latent_k, self._var_list))}

    def _run(self, feed_dict):
        bind_feed(feed_dict, self._var_list[0].device)
        self._update(self._var_list, self._grad_func, feed_dict,
                     _capi.current_stream())
        # (the latents were written through the C-ABI: a sampler that keeps
        # something about the same tensors must see it)
        _writes.note(self._var_list)
        self.t += 1                                           # sgmcmc.py:106

    def _define_variables(self, qs):
        raise NotImplementedError()

    def _update(self, qs, grad_func, feed_dict, stream):
        raise NotImplementedError()


class SGLD(SGMCMC):
    """Stochastic Gradient Langevin Dynamics (sgmcmc.py:169-204):
    q <- q + lr/2 * grad + N(0, lr)."""

    def __init__(self, learning_rate, seed=None):
        self.lr = learning_rate
        super(SGLD, self).__init__(seed)

    def _define_variables(self, qs):
        self._aux = [None] * len(qs)

    def _hps(self):
        return 0.0, 0.0

    def _update(self, qs, grad_func, feed_dict, stream):
        lr = _value(self.lr, feed_dict, 'learning_rate')
        decay, eps = self._hps()
        grads = grad_func(qs)
        for k, (q, g, aux) in enumerate(zip(qs, grads, self._aux)):
            _capi.call('zshmc_sgld_update', q.data_ptr(), g.data_ptr(),
                       _capi.ptr(aux), lr, decay, eps, q.numel(), self.seed,
                       self.t & 0xFFFFFFFF, k, stream)


class PSGLD(SGLD):
    """Preconditioned SGLD with the RMSprop preconditioner
    (sgmcmc.py:207-253; `preconditioner_hparams` = RMSHParams(decay, epsilon),
    default (0.9, 1e-3))."""

    class RMSPreconditioner:
        HParams = namedtuple('RMSHParams', 'decay epsilon')
        default_hps = HParams(decay=0.9, epsilon=1e-3)

    def __init__(self, learning_rate, preconditioner='rms',
                 preconditioner_hparams=None, seed=None):
        self.preconditioner = {'rms': PSGLD.RMSPreconditioner}[preconditioner]
        if preconditioner_hparams is None:
            preconditioner_hparams = self.preconditioner.default_hps
        self.preconditioner_hparams = preconditioner_hparams
        super(PSGLD, self).__init__(learning_rate, seed)

    def _define_variables(self, qs):
        self.vs = [torch.zeros_like(q) for q in qs]           # sgmcmc.py:229-230
        self._aux = self.vs

    def _hps(self):
        h = self.preconditioner_hparams
        return float(h.decay), float(h.epsilon)


class SGHMC(SGMCMC):
    """Stochastic Gradient HMC (sgmcmc.py:256-363), first- or second-order
    integrator; `sgmcmc_info.mean_k[name]` is the mean kinetic energy
    mean(v'^2) of the updated momentum."""

    def __init__(self, learning_rate, friction=0.25, variance_estimate=0.,
                 n_iter_resample_v=20, second_order=True, seed=None):
        self.lr = learning_rate
        self.alpha = friction
        self.beta = variance_estimate
        self.n_iter_resample_v = 0 if n_iter_resample_v is None else \
            n_iter_resample_v
        self.second_order = bool(second_order)
        super(SGHMC, self).__init__(seed)

    def _init_momentum(self, lr0):
        for k, v in enumerate(self.vs):
            _capi.call('zshmc_sg_momentum', v.data_ptr(), math.sqrt(lr0),
                       v.numel(), self.seed, _INIT_ITER, k,
                       _capi.current_stream())
        self._v_ready = True

    def _define_variables(self, qs):
        dev = qs[0].device
        self.vs = [torch.empty_like(q) for q in qs]
        # v0 ~ N(0, lr) (sgmcmc.py:310-314).  A learning-rate placeholder
        # without a default has no value yet: in the reference the variable
        # initialiser would have to be fed too; here the draw waits for the
        # first run's value.
        self._v_ready = False
        if not isinstance(self.lr, placeholder) or self.lr.default is not None:
            self._init_momentum(_value(self.lr, None, 'learning_rate'))
        # {sum v_old^2, sum v'^2} per latent, and the scalar mean_k outputs
        self._sums = [torch.zeros(2, dtype=torch.float64, device=dev)
                      for _ in qs]
        self._mean_k = [torch.zeros(1, device=dev) for _ in qs]
        self._dummy_alpha = [torch.zeros(2, device=dev) for _ in qs]

    def _info_dicts(self):
        d = super(SGHMC, self)._info_dicts()
        d['mean_k'] = dict(zip(self._latent_k,
                               [m[0] for m in self._mean_k]))
        return d

    def _resample(self, lr, feed_dict, stream):
        if not self._v_ready:
            self._init_momentum(lr)
        n = int(_value(self.n_iter_resample_v, feed_dict, 'n_iter_resample_v'))
        if n != 0 and self.t % n == 0:                        # sgmcmc.py:319-326
            for k, v in enumerate(self.vs):
                _capi.call('zshmc_sg_momentum', v.data_ptr(), math.sqrt(lr),
                           v.numel(), self.seed, self.t & 0xFFFFFFFF, k,
                           stream)

    def _update(self, qs, grad_func, feed_dict, stream):
        lr = _value(self.lr, feed_dict, 'learning_rate')
        alpha = _value(self.alpha, feed_dict, 'friction')
        beta = _value(self.beta, feed_dict, 'variance_estimate')
        self._resample(lr, feed_dict, stream)
        noise_std = math.sqrt(max(2.0 * (alpha - beta) * lr, 0.0))
        if self.second_order:                                 # q1 = q + v/2
            for q, v in zip(qs, self.vs):
                _capi.call('zshmc_sg_half_drift', q.data_ptr(), v.data_ptr(),
                           q.numel(), None, stream)
        grads = grad_func(qs)
        for k, (q, v, g) in enumerate(zip(qs, self.vs, grads)):
            sums = self._sums[k]
            _capi.call('zshmc_sghmc_update', q.data_ptr(), v.data_ptr(),
                       g.data_ptr(), q.numel(), lr, alpha, noise_std,
                       int(self.second_order), self.seed, self.t & 0xFFFFFFFF,
                       k, sums.data_ptr() + 8, stream)
            # mean_k = sum/n on the device (tune_rate 0: alpha untouched)
            _capi.call('zshmc_sgnht_scalar', self._dummy_alpha[k].data_ptr(),
                       sums.data_ptr(), q.numel(), lr, 0.0,
                       int(self.second_order), 1, self._mean_k[k].data_ptr(),
                       stream)


class SGNHT(SGHMC):
    """Stochastic Gradient Nose-Hoover Thermostat (sgmcmc.py:366-497) with a
    vector (per element) or scalar friction; `sgmcmc_info.alpha[name]` and
    `.mean_k[name]` follow `use_vector_alpha`."""

    def __init__(self, learning_rate, variance_extra=0., tune_rate=1.,
                 n_iter_resample_v=None, second_order=True,
                 use_vector_alpha=True, seed=None):
        self.lr = learning_rate
        self.a = variance_extra
        self.tune_rate = tune_rate
        self.n_iter_resample_v = 0 if n_iter_resample_v is None else \
            n_iter_resample_v
        self.second_order = bool(second_order)
        self.use_vector_alpha = bool(use_vector_alpha)
        SGMCMC.__init__(self, seed)

    def _define_variables(self, qs):
        SGHMC._define_variables(self, qs)
        a0 = _value(self.a, None, 'variance_extra')
        dev = qs[0].device
        if self.use_vector_alpha:                             # sgmcmc.py:448-450
            self.alphas = [torch.full_like(q, a0) for q in qs]
            self._mean_k = [torch.zeros_like(q) for q in qs]
        else:
            # {alpha, alpha of the step being integrated}
            self._alpha_s = [torch.full((2,), a0, device=dev) for _ in qs]
            self.alphas = [a[0] for a in self._alpha_s]

    def _info_dicts(self):
        d = SGMCMC._info_dicts(self)
        if self.use_vector_alpha:
            d['mean_k'] = dict(zip(self._latent_k, self._mean_k))
        else:
            d['mean_k'] = dict(zip(self._latent_k,
                                   [m[0] for m in self._mean_k]))
        d['alpha'] = dict(zip(self._latent_k, self.alphas))
        return d

    def _update(self, qs, grad_func, feed_dict, stream):
        lr = _value(self.lr, feed_dict, 'learning_rate')
        a = _value(self.a, feed_dict, 'variance_extra')
        tune = _value(self.tune_rate, feed_dict, 'tune_rate')
        so = int(self.second_order)
        self._resample(lr, feed_dict, stream)
        noise_std = math.sqrt(max(2.0 * a * lr, 0.0))
        scalar = not self.use_vector_alpha
        if self.second_order:
            for k, (q, v) in enumerate(zip(qs, self.vs)):
                _capi.call('zshmc_sg_half_drift', q.data_ptr(), v.data_ptr(),
                           q.numel(),
                           self._sums[k].data_ptr() if scalar else None,
                           stream)
        if scalar:
            for k, q in enumerate(qs):
                _capi.call('zshmc_sgnht_scalar', self._alpha_s[k].data_ptr(),
                           self._sums[k].data_ptr(), q.numel(), lr, tune, so,
                           0, None, stream)
        grads = grad_func(qs)
        for k, (q, v, g) in enumerate(zip(qs, self.vs, grads)):
            if scalar:
                _capi.call('zshmc_sgnht_update', q.data_ptr(), v.data_ptr(),
                           g.data_ptr(), None, self._alpha_s[k].data_ptr(),
                           None, q.numel(), lr, tune, noise_std, so, self.seed,
                           self.t & 0xFFFFFFFF, k,
                           self._sums[k].data_ptr() + 8, stream)
                _capi.call('zshmc_sgnht_scalar', self._alpha_s[k].data_ptr(),
                           self._sums[k].data_ptr(), q.numel(), lr, tune, so,
                           1, self._mean_k[k].data_ptr(), stream)
            else:
                _capi.call('zshmc_sgnht_update', q.data_ptr(), v.data_ptr(),
                           g.data_ptr(), self.alphas[k].data_ptr(), None,
                           self._mean_k[k].data_ptr(), q.numel(), lr, tune,
                           noise_std, so, self.seed, self.t & 0xFFFFFFFF, k,
                           None, stream)
