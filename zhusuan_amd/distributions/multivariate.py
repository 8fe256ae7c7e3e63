"""UnnormalizedMultinomial -- on the HMC path only because the
logistic-normal topic model needs it (reference
examples/topic_models/lntm_mcem.py:46).  Mirrors
zhusuan/distributions/multivariate.py:339-449."""
import torch

from .. import _ops
from .base import Distribution, as_tensor, common_device, default_device
from .univariate import _assert_same_float_dtype, _require_f32, _FLOATS, _INTS

__all__ = ['UnnormalizedMultinomial', 'BagofCategoricals', 'LogMixture',
           'log_mixture']


class LogMixture(object):
    """Lazy `log(theta @ phi)`: the logits [..., V] of the logistic-normal
    topic model (lntm_mcem.py:41-46) that are never materialised.
    `UnnormalizedMultinomial(log_mixture(theta, phi), normalize_logits=False)`
    evaluates log_prob and d/dtheta with the fused fp32-MFMA kernel
    (csrc/linear_bernoulli.hip, multinomial mode); anything else falls back to
    `.dense()`."""

    def __init__(self, theta, phi):
        theta, phi = as_tensor(theta), as_tensor(phi)
        if phi.dim() != 2 or theta.dim() < 1 or theta.shape[-1] != phi.shape[0]:
            raise ValueError(
                "log_mixture: theta[..., K] and phi[K, V] expected, got {} and {}"
                .format(tuple(theta.shape), tuple(phi.shape)))
        self.theta, self.phi = theta, phi

    @property
    def shape(self):
        return torch.Size(tuple(self.theta.shape[:-1]) + (self.phi.shape[1],))

    @property
    def dtype(self):
        return self.theta.dtype

    @property
    def device(self):
        return self.theta.device

    def dim(self):
        return self.theta.dim()

    def dense(self):
        return torch.log(self.theta @ self.phi)


def log_mixture(theta, phi):
    return LogMixture(theta, phi)


class UnnormalizedMultinomial(Distribution):
    def __init__(self, logits, normalize_logits=True, dtype=torch.int32,
                 group_ndims=0, **kwargs):
        self._lazy = None
        if isinstance(logits, LogMixture):
            if logits.dtype != torch.float32:
                raise TypeError("UnnormalizedMultinomial: log_mixture must be "
                                "float32")
            # fused only without re-normalisation and without a gradient
            # through phi; otherwise the dense logits
            if normalize_logits or logits.phi.requires_grad or \
                    logits.phi.shape[0] > 256:
                logits = logits.dense()
            else:
                self._lazy = logits
        if self._lazy is not None:
            if dtype not in _FLOATS + _INTS:
                raise TypeError(
                    "`dtype`({}) must be int or float.".format(dtype))
            self._logits = None
            self._n_categories = int(self._lazy.shape[-1])
            self.normalize_logits = normalize_logits
            super(UnnormalizedMultinomial, self).__init__(
                dtype=dtype, param_dtype=torch.float32, is_continuous=False,
                is_reparameterized=False, group_ndims=group_ndims, **kwargs)
            return
        dev = common_device(logits) or default_device()
        self._logits = as_tensor(logits, dtype=None if isinstance(
            logits, torch.Tensor) else torch.float32, device=dev)
        param_dtype = _assert_same_float_dtype(
            [(self._logits, 'UnnormalizedMultinomial.logits')])
        _require_f32(param_dtype, 'UnnormalizedMultinomial')
        if dtype not in _FLOATS + _INTS:
            raise TypeError("`dtype`({}) must be int or float.".format(dtype))
        if self._logits.dim() < 1:
            raise ValueError("UnnormalizedMultinomial.logits should have "
                             "rank >= 1, got a scalar.")
        self._n_categories = int(self._logits.shape[-1])
        self.normalize_logits = normalize_logits
        super(UnnormalizedMultinomial, self).__init__(
            dtype=dtype, param_dtype=param_dtype, is_continuous=False,
            is_reparameterized=False, group_ndims=group_ndims, **kwargs)

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._lazy.dense()
        return self._logits

    @property
    def n_categories(self):
        return self._n_categories

    def _device(self):
        return self._lazy.device if self._lazy is not None \
            else self._logits.device

    def _get_value_shape(self):
        return torch.Size([self._n_categories])

    def _get_batch_shape(self):
        return self._lazy.shape[:-1] if self._lazy is not None \
            else self._logits.shape[:-1]

    def _sample(self, n_samples):
        raise NotImplementedError(
            "Unnormalized multinomial distribution does not support sampling "
            "because n_experiments is not given. Please use class "
            "Multinomial to sample")

    def _log_prob(self, given):
        given = given.to(self.param_dtype)          # :436
        lazy = self._lazy
        if lazy is not None:
            batch = tuple(lazy.shape[:-1])
            gs = tuple(given.shape)
            n_cat = self._n_categories
            rows = 1
            for d in batch:
                rows *= int(d)
            # counts shared by the leading (chain) axes: [*batch_tail, V]
            ok = (len(gs) >= 1 and gs[-1] == n_cat and len(gs) - 1 <= len(batch)
                  and gs[:-1] == batch[len(batch) - (len(gs) - 1):])
            if ok and rows > 0:
                return _ops.MixtureMultinomialLogLik.apply(
                    lazy.theta, lazy.phi, given.reshape(-1, n_cat))
            self._logits = lazy.dense()
        try:
            full = torch.broadcast_shapes(given.shape, self.logits.shape)
        except RuntimeError:
            raise ValueError(
                "given and logits cannot broadcast to match. ({} vs. {})"
                .format(tuple(given.shape), tuple(self.logits.shape)))
        n_cat = self._n_categories
        g = given.expand(full).contiguous().reshape(-1, n_cat)
        l = self.logits.expand(full).contiguous().reshape(-1, n_cat)
        out = _ops.UnnormalizedMultinomialLogProb.apply(
            l, g, self.normalize_logits)
        return out.reshape(full[:-1])


BagofCategoricals = UnnormalizedMultinomial
