"""Distribution base class: batch/value shapes, the `group_ndims` reduction
and the sample(n_samples) squeeze rule.  Mirrors reference
zhusuan/distributions/base.py:17-332 on torch device tensors."""
import torch

from .. import _symbolic

from ..utils import broadcast_shapes

__all__ = ['Distribution']


def as_tensor(value, dtype=None, device=None, keep_symbolic=False):
    """tf.convert_to_tensor analogue.  Python / NumPy values become tensors
    on `device` (default: current HIP device when available).  A symbolic
    latent expression (zhusuan_amd/_symbolic.py) is replaced by its value
    unless the caller keeps it symbolic (an observed latent stays a symbol so
    that the model function's own ops on it can be recognised)."""
    if isinstance(value, torch.Tensor):
        if isinstance(value, _symbolic.Sym) and not (
                keep_symbolic and (dtype is None or value.dtype == dtype)):
            value = value.force()
        t = value
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t
    if hasattr(value, 'tensor') and not isinstance(value, (list, tuple)):
        # StochasticTensor -> its current value (bn.py:164-175)
        return as_tensor(value.tensor, dtype, device, keep_symbolic)
    if device is None:
        device = default_device()
    return torch.as_tensor(value, dtype=dtype, device=device)


def default_device():
    return torch.device('cuda', torch.cuda.current_device()) \
        if torch.cuda.is_available() else torch.device('cpu')


def common_device(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return None


class Distribution(object):
    """base.py:17-120."""

    # read-only attributes of the reference's Distribution (base.py:123-181),
    # filled by the constructor and exposed as properties below the class
    _FIELDS = ('dtype', 'param_dtype', 'is_continuous', 'is_reparameterized',
               'use_path_derivative', 'group_ndims')

    def __init__(self, dtype, param_dtype, is_continuous, is_reparameterized,
                 use_path_derivative=False, group_ndims=0, **kwargs):
        if isinstance(group_ndims, bool) or not isinstance(group_ndims, int):
            raise TypeError("group_ndims should be a Python int, got {!r}"
                            .format(group_ndims))
        if group_ndims < 0:
            raise ValueError("group_ndims must be non-negative.")
        for field, value in zip(self._FIELDS, (
                dtype, param_dtype, is_continuous, is_reparameterized,
                use_path_derivative, group_ndims)):
            setattr(self, '_' + field, value)

    def path_param(self, param):
        """base.py:183-190: stop gradients through parameters when the path
        derivative estimator is requested."""
        return param.detach() if self._use_path_derivative else param

    # -- shapes: torch shapes are always concrete, so the reference's static
    # (`get_*_shape()`) and dynamic (`*_shape`) views are one and the same ----
    def get_value_shape(self):
        return self._get_value_shape()

    def get_batch_shape(self):
        return self._get_batch_shape()

    value_shape = property(lambda self: self._get_value_shape())
    batch_shape = property(lambda self: self._get_batch_shape())

    def _get_value_shape(self):
        raise NotImplementedError()

    def _get_batch_shape(self):
        raise NotImplementedError()

    # -- sampling, base.py:236-263 ----------------------------------------
    def sample(self, n_samples=None):
        if n_samples is None:
            return self._sample(n_samples=1).squeeze(0)
        if isinstance(n_samples, torch.Tensor):
            if n_samples.dim() != 0:
                raise ValueError(
                    "n_samples should be a scalar (0-D Tensor).")
            n_samples = int(n_samples.item())
        return self._sample(int(n_samples))

    def _sample(self, n_samples):
        raise NotImplementedError()

    # -- densities, base.py:271-320 ---------------------------------------
    def _check_input_shape(self, given):
        given = as_tensor(given, dtype=self.dtype,
                          device=self._device())
        err_msg = "The given argument should be able to broadcast to " \
                  "match batch_shape + value_shape of the distribution."
        sample_shape = tuple(self.get_batch_shape()) + tuple(
            self.get_value_shape())
        try:
            broadcast_shapes(tuple(given.shape), sample_shape)
        except RuntimeError:
            raise ValueError(
                err_msg + " ({} vs. {} + {})".format(
                    tuple(given.shape), tuple(self.get_batch_shape()),
                    tuple(self.get_value_shape())))
        return given

    def _device(self):
        return None

    def log_prob(self, given):
        """log density (mass) at `given`, summed over the last `group_ndims`
        batch axes (base.py:290-304)."""
        given = self._check_input_shape(given)
        return self._log_prob_grouped(given)

    def _log_prob_grouped(self, given):
        """Default: element-wise kernel then a trailing-axes sum.  Subclasses
        whose kernel fuses the reduction override this."""
        log_p = self._log_prob(given)
        if self._group_ndims == 0:
            return log_p
        if self._group_ndims > log_p.dim():
            raise ValueError("group_ndims {} exceeds log_prob rank {}"
                             .format(self._group_ndims, log_p.dim()))
        return log_p.sum(dim=tuple(range(-self._group_ndims, 0)))

    def prob(self, given):
        """base.py:306-320."""
        return torch.exp(self.log_prob(given))

    def _log_prob(self, given):
        raise NotImplementedError()


for _f in Distribution._FIELDS:
    setattr(Distribution, _f, property(
        lambda self, _k='_' + _f: getattr(self, _k),
        doc="The reference's Distribution.%s (read-only)." % _f))
