"""Multivariate distributions on the HMC path: UnnormalizedMultinomial (the
logistic-normal topic model needs it, reference
examples/topic_models/lntm_mcem.py:46; zhusuan/distributions/
multivariate.py:339-449) and MultivariateNormalCholesky (SURVEY 8f-4;
multivariate.py:41-193)."""
import torch

from ..utils import broadcast_shapes

from .. import _capi, _ops, _symbolic
from ..utils import next_op_offset
from .base import Distribution, as_tensor, common_device, default_device
from .univariate import _assert_same_float_dtype, _require_f32, _FLOATS, _INTS

__all__ = ['UnnormalizedMultinomial', 'BagofCategoricals', 'LogMixture',
           'log_mixture', 'MultivariateNormalCholesky']


class MultivariateNormalCholesky(Distribution):
    """Multivariate normal with covariance L L^T given by its lower-triangular
    Cholesky factor (multivariate.py:41-193; same arguments, shape rules and
    error messages).  `mean [..., n_dim]`, `cov_tril [..., n_dim, n_dim]`;
    log_prob, its gradients and sampling run in csrc/mvn.hip."""

    def __init__(self, mean, cov_tril, group_ndims=0, is_reparameterized=True,
                 use_path_derivative=False, check_numerics=False, **kwargs):
        self._check_numerics = check_numerics
        dev = common_device(mean, cov_tril) or default_device()
        f32 = torch.float32
        self._mean = as_tensor(mean, dtype=None if isinstance(
            mean, torch.Tensor) else f32, device=dev)
        if self._mean.dim() < 1:                              # :83-84
            raise ValueError("MultivariateNormalCholesky.mean should have "
                             "rank >= 1, got a scalar.")
        self._n_dim = int(self._mean.shape[-1])
        self._cov_tril = as_tensor(cov_tril, dtype=None if isinstance(
            cov_tril, torch.Tensor) else f32, device=dev)
        if self._cov_tril.dim() < 2:                          # :87-88
            raise ValueError("MultivariateNormalCholesky.cov_tril should "
                             "have rank >= 2, got rank {}."
                             .format(self._cov_tril.dim()))
        expected = tuple(self._mean.shape) + (self._n_dim,)   # :90-103
        if tuple(self._cov_tril.shape) != expected:
            raise ValueError(
                "MultivariateNormalCholesky.cov_tril should have compatible "
                "shape with mean. Expected {} got {}".format(
                    expected, tuple(self._cov_tril.shape)))
        dtype = _assert_same_float_dtype(
            [(self._mean, 'MultivariateNormalCholesky.mean'),
             (self._cov_tril, 'MultivariateNormalCholesky.cov_tril')])
        _require_f32(dtype, 'MultivariateNormalCholesky')
        if self._n_dim > 512:
            raise ValueError("MultivariateNormalCholesky: n_dim <= 512 on "
                             "this device path, got {}".format(self._n_dim))
        super(MultivariateNormalCholesky, self).__init__(
            dtype=dtype, param_dtype=dtype, is_continuous=True,
            is_reparameterized=is_reparameterized,
            use_path_derivative=use_path_derivative,
            group_ndims=group_ndims, **kwargs)

    @property
    def mean(self):
        return self._mean

    @property
    def cov_tril(self):
        return self._cov_tril

    def _device(self):
        return self._mean.device

    def _get_value_shape(self):
        return torch.Size([self._n_dim])

    def _get_batch_shape(self):
        return self._mean.shape[:-1]

    def _sample(self, n_samples):
        """multivariate.py:141-164: mean + L . N(0, I)."""
        mean, tril = self._mean, self._cov_tril
        if not self.is_reparameterized:
            mean, tril = mean.detach(), tril.detach()
        _ops.require_device(mean, tril)
        n, D = int(n_samples), self._n_dim
        batch = tuple(self._get_batch_shape())
        b = 1
        for d in batch:
            b *= int(d)
        shape = (n,) + batch + (D,)
        seed, offset = next_op_offset()
        stream = _capi.current_stream()
        if mean.requires_grad or tril.requires_grad:
            # reparameterisation: draw the N(0, I) block with the kernel and
            # let autograd see  mean + L . eps
            eps = torch.empty(shape, dtype=torch.float32, device=mean.device)
            one = torch.ones(1, device=mean.device)
            zero = torch.zeros(1, device=mean.device)
            _capi.call('zshmc_normal_sample', eps.data_ptr(), zero.data_ptr(),
                       one.data_ptr(), eps.numel(), 1, _capi.BCAST_SCALAR,
                       _capi.BCAST_SCALAR, seed, offset, stream)
            return mean + (tril @ eps.unsqueeze(-1)).squeeze(-1)
        out = torch.empty(shape, dtype=torch.float32, device=mean.device)
        m, t = mean.detach().contiguous(), tril.detach().contiguous()
        _capi.call('zshmc_mvn_tril_sample', out.data_ptr(), m.data_ptr(),
                   t.data_ptr(), n * b, D, max(b, 1), max(b, 1), seed, offset,
                   stream)
        return out

    def _log_prob(self, given):
        mean = self.path_param(self._mean)
        tril = self.path_param(self._cov_tril)
        full = broadcast_shapes(given.shape, mean.shape)
        x = given if tuple(given.shape) == tuple(full) else given.expand(full)
        mb = tuple(mean.shape[:-1])
        if len(mb) and tuple(full[len(full) - 1 - len(mb):-1]) != mb:
            # `given` broadcasts INTO the batch axes: materialise parameters
            mean = mean.expand(full)
            tril = tril.expand(tuple(full) + (self._n_dim,))
        out = _ops.MvnTrilLogProb.apply(x, mean, tril)
        if self._check_numerics and not bool(torch.isfinite(out).all()):
            raise FloatingPointError(
                "MultivariateNormalCholesky.log_prob: Tensor had Inf or NaN")
        return out


class LogMixture(object):
    """Lazy `log(theta @ phi)`: the logits [..., V] of the logistic-normal
    topic model (lntm_mcem.py:41-46) that are never materialised.
    `UnnormalizedMultinomial(log_mixture(theta, phi), normalize_logits=False)`
    evaluates log_prob and d/dtheta with the fused fp32-MFMA kernel
    (csrc/linear_bernoulli.hip, multinomial mode); anything else falls back to
    `.dense()`."""

    def __init__(self, theta, phi):
        src = _symbolic.softmax_source(theta)
        if src is not None:
            # theta = softmax(latent) still symbolic: formed only on demand
            made = LogMixture.of_softmax(src, phi, tuple(theta.shape[:-1]))
            self.__dict__.update(made.__dict__)
            return
        theta, phi = as_tensor(theta), as_tensor(phi)
        if phi.dim() != 2 or theta.dim() < 1 or theta.shape[-1] != phi.shape[0]:
            raise ValueError(
                "log_mixture: theta[..., K] and phi[K, V] expected, got {} and {}"
                .format(tuple(theta.shape), tuple(phi.shape)))
        self._theta, self.phi = theta, phi
        self.softmax_source = None
        self._batch = tuple(theta.shape[:-1])
        self._dtype, self._device = theta.dtype, theta.device

    @classmethod
    def of_softmax(cls, source, phi, batch_shape):
        """`log(softmax(source, -1).reshape(batch_shape + [K]) @ phi)` with
        theta formed only if somebody asks (the native HMC plan computes the
        softmax inside its own element-wise launch): what the literal
        spelling `torch.log(torch.softmax(eta, -1) @ phi)` lowers to
        (zhusuan_amd/_symbolic.py)."""
        phi = as_tensor(phi)
        k = int(source.shape[-1])
        rows = 1
        for d in batch_shape:
            rows *= int(d)
        if phi.dim() != 2 or phi.shape[0] != k or \
                rows * k != source.numel():
            raise ValueError(
                "log_mixture: softmax source {} and phi {} do not fit batch "
                "shape {}".format(tuple(source.shape), tuple(phi.shape),
                                  tuple(batch_shape)))
        self = cls.__new__(cls)
        self._theta, self.phi = None, phi
        self.softmax_source = source
        self._batch = tuple(int(d) for d in batch_shape)
        self._dtype, self._device = source.dtype, source.device
        return self

    @property
    def theta(self):
        if self._theta is None:
            src = self.softmax_source
            self._theta = torch.softmax(src, -1).reshape(
                self._batch + (int(src.shape[-1]),))
        return self._theta

    @property
    def shape(self):
        return torch.Size(self._batch + (self.phi.shape[1],))

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def dim(self):
        return len(self._batch) + 1

    def dense(self):
        return torch.log(self.theta @ self.phi)


def log_mixture(theta, phi):
    return LogMixture(theta, phi)


class UnnormalizedMultinomial(Distribution):
    def __init__(self, logits, normalize_logits=True, dtype=torch.int32,
                 group_ndims=0, **kwargs):
        self._lazy = None
        logits = _symbolic.lower_multinomial_logits(logits)
        if isinstance(logits, LogMixture):
            if logits.dtype != torch.float32:
                raise TypeError("UnnormalizedMultinomial: log_mixture must be "
                                "float32")
            # fused only without re-normalisation and without a gradient
            # through phi; otherwise the dense logits
            if normalize_logits or logits.phi.requires_grad or \
                    logits.phi.shape[0] > _ops.MAX_LIKELIHOOD_WIDTH:
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
            full = broadcast_shapes(given.shape, self.logits.shape)
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
