"""UnnormalizedMultinomial -- on the HMC path only because the
logistic-normal topic model needs it (reference
examples/topic_models/lntm_mcem.py:46).  Mirrors
zhusuan/distributions/multivariate.py:339-449."""
import torch

from .. import _ops
from .base import Distribution, as_tensor, common_device, default_device
from .univariate import _assert_same_float_dtype, _require_f32, _FLOATS, _INTS

__all__ = ['UnnormalizedMultinomial', 'BagofCategoricals']


class UnnormalizedMultinomial(Distribution):
    def __init__(self, logits, normalize_logits=True, dtype=torch.int32,
                 group_ndims=0, **kwargs):
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
        return self._logits

    @property
    def n_categories(self):
        return self._n_categories

    def _device(self):
        return self._logits.device

    def _get_value_shape(self):
        return torch.Size([self._n_categories])

    def _get_batch_shape(self):
        return self._logits.shape[:-1]

    def _sample(self, n_samples):
        raise NotImplementedError(
            "Unnormalized multinomial distribution does not support sampling "
            "because n_experiments is not given. Please use class "
            "Multinomial to sample")

    def _log_prob(self, given):
        given = given.to(self.param_dtype)          # :436
        try:
            full = torch.broadcast_shapes(given.shape, self._logits.shape)
        except RuntimeError:
            raise ValueError(
                "given and logits cannot broadcast to match. ({} vs. {})"
                .format(tuple(given.shape), tuple(self._logits.shape)))
        n_cat = self._n_categories
        g = given.expand(full).contiguous().reshape(-1, n_cat)
        l = self._logits.expand(full).contiguous().reshape(-1, n_cat)
        out = _ops.UnnormalizedMultinomialLogProb.apply(
            l, g, self.normalize_logits)
        return out.reshape(full[:-1])


BagofCategoricals = UnnormalizedMultinomial
