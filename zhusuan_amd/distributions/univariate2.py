"""Laplace / Gamma / InverseGamma / Beta: the
remaining two-parameter continuous families of reference
zhusuan/distributions/univariate.py (:1164-1277, :662-751, :1070-1161,
:753-855) with the same constructor contracts, error messages and shape
rules; log_prob, its analytic gradients and sampling run in the HIP kernels
of csrc/distributions2.hip."""
import torch

from ..utils import broadcast_shapes

from .. import _capi, _ops
from ..utils import next_op_offset
from .base import Distribution, as_tensor, common_device, default_device
from .univariate import _assert_same_float_dtype, _require_f32

__all__ = ['Laplace', 'Gamma', 'InverseGamma', 'Beta']


class _TwoParam(Distribution):
    """Shared plumbing: two broadcastable float32 parameters."""
    _kind = None
    _names = ('a', 'b')

    def _setup(self, a, b, group_ndims, check_numerics, is_reparameterized,
               use_path_derivative=False, **kwargs):
        cls = type(self).__name__
        dev = common_device(a, b) or default_device()
        f32 = torch.float32
        self._a = as_tensor(a, dtype=None if isinstance(a, torch.Tensor)
                            else f32, device=dev)
        self._b = as_tensor(b, dtype=None if isinstance(b, torch.Tensor)
                            else f32, device=dev)
        dtype = _assert_same_float_dtype(
            [(self._a, '%s.%s' % (cls, self._names[0])),
             (self._b, '%s.%s' % (cls, self._names[1]))])
        _require_f32(dtype, cls)
        try:
            broadcast_shapes(self._a.shape, self._b.shape)
        except RuntimeError:
            raise ValueError(
                "{} and {} should be broadcastable to match each "
                "other. ({} vs. {})".format(
                    self._names[0], self._names[1], tuple(self._a.shape),
                    tuple(self._b.shape)))
        self._check_numerics = check_numerics
        super(_TwoParam, self).__init__(
            dtype=dtype, param_dtype=dtype, is_continuous=True,
            is_reparameterized=is_reparameterized,
            use_path_derivative=use_path_derivative,
            group_ndims=group_ndims, **kwargs)

    def _device(self):
        return self._a.device

    def _get_value_shape(self):
        return torch.Size([])

    def _get_batch_shape(self):
        return broadcast_shapes(self._a.shape, self._b.shape)

    def _params_for_log_prob(self):
        return self._a, self._b

    def _log_prob_grouped(self, given):
        a, b = self._params_for_log_prob()
        full = broadcast_shapes(given.shape, a.shape, b.shape)
        if self._group_ndims > len(full):
            raise ValueError("group_ndims {} exceeds log_prob rank {}"
                             .format(self._group_ndims, len(full)))
        out = _ops.Uni2LogProb.apply(given, a, b, self._kind,
                                     self._group_ndims)
        if self._check_numerics and not bool(torch.isfinite(out).all()):
            raise FloatingPointError(
                "%s.log_prob : Tensor had Inf or NaN" % type(self).__name__)
        return out

    def _log_prob(self, given):
        a, b = self._params_for_log_prob()
        return _ops.Uni2LogProb.apply(given, a, b, self._kind, 0)

    def _draw(self, n_samples, a, b):
        batch = self._get_batch_shape()
        ae = a.detach().expand(batch).contiguous() if a.numel() != 1 else a.detach()
        be = b.detach().expand(batch).contiguous() if b.numel() != 1 else b.detach()
        _ops.require_device(ae, be)
        inner = 1
        for d in batch:
            inner *= int(d)
        n = int(n_samples) * inner
        out = torch.empty((int(n_samples),) + tuple(batch),
                          dtype=torch.float32, device=ae.device)
        seed, offset = next_op_offset()
        _capi.call('zshmc_uni2_sample', self._kind, out.data_ptr(),
                   ae.data_ptr(), be.data_ptr(), n, max(inner, 1),
                   _capi.BCAST_SCALAR if ae.numel() == 1 else _capi.BCAST_FULL,
                   _capi.BCAST_SCALAR if be.numel() == 1 else _capi.BCAST_FULL,
                   seed, offset, _capi.current_stream())
        return out

    def _sample(self, n_samples):
        return self._draw(n_samples, self._a, self._b)


class Laplace(_TwoParam):
    """Univariate Laplace (univariate.py:1164-1277)."""
    _kind = 0
    _names = ('loc', 'scale')

    def __init__(self, loc, scale, group_ndims=0, is_reparameterized=True,
                 use_path_derivative=False, check_numerics=False, **kwargs):
        self._setup(loc, scale, group_ndims, check_numerics,
                    is_reparameterized, use_path_derivative, **kwargs)

    @property
    def loc(self):
        return self._a

    @property
    def scale(self):
        return self._b

    def _params_for_log_prob(self):
        return self.path_param(self._a), self.path_param(self._b)

    def _sample(self, n_samples):
        """loc - scale * sign(u) * log1p(-|u|), u in (-1, 1) (:1246-1265).
        With is_reparameterized the standard draw comes from the kernel and
        autograd sees `loc + scale * eps`."""
        if self.is_reparameterized and (self._a.requires_grad or
                                        self._b.requires_grad):
            zero = torch.zeros(1, device=self._a.device)
            one = torch.ones(1, device=self._a.device)
            batch = self._get_batch_shape()
            inner = 1
            for d in batch:
                inner *= int(d)
            eps = torch.empty((int(n_samples),) + tuple(batch),
                              dtype=torch.float32, device=self._a.device)
            seed, offset = next_op_offset()
            _capi.call('zshmc_uni2_sample', 0, eps.data_ptr(), zero.data_ptr(),
                       one.data_ptr(), int(n_samples) * inner, max(inner, 1),
                       _capi.BCAST_SCALAR, _capi.BCAST_SCALAR, seed, offset,
                       _capi.current_stream())
            return self._a + self._b * eps
        return self._draw(n_samples, self._a, self._b)


class Gamma(_TwoParam):
    """Univariate Gamma, rate parameterisation (univariate.py:662-751)."""
    _kind = 1
    _names = ('alpha', 'beta')

    def __init__(self, alpha, beta, group_ndims=0, check_numerics=False,
                 **kwargs):
        self._setup(alpha, beta, group_ndims, check_numerics, False, **kwargs)

    @property
    def alpha(self):
        return self._a

    @property
    def beta(self):
        return self._b


class InverseGamma(_TwoParam):
    """Univariate inverse Gamma (univariate.py:1070-1161)."""
    _kind = 2
    _names = ('alpha', 'beta')

    def __init__(self, alpha, beta, group_ndims=0, check_numerics=False,
                 **kwargs):
        self._setup(alpha, beta, group_ndims, check_numerics, False, **kwargs)

    @property
    def alpha(self):
        return self._a

    @property
    def beta(self):
        return self._b


class Beta(_TwoParam):
    """Univariate Beta (univariate.py:753-855)."""
    _kind = 3
    _names = ('alpha', 'beta')

    def __init__(self, alpha, beta, group_ndims=0, check_numerics=False,
                 **kwargs):
        self._setup(alpha, beta, group_ndims, check_numerics, False, **kwargs)

    @property
    def alpha(self):
        return self._a

    @property
    def beta(self):
        return self._b
