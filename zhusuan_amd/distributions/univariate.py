"""Normal / Bernoulli / Categorical on the HMC hot path.  Mirrors the
constructor contracts, error messages and shape rules of reference
zhusuan/distributions/univariate.py:43-184 (Normal), :334-406 (Bernoulli),
:409-551 (Categorical); all arithmetic runs in the HIP kernels of
csrc/distributions.hip through zhusuan_amd._ops."""
import torch

from ..utils import broadcast_shapes

from .. import _capi, _ops, _symbolic
from ..utils import next_op_offset
from .base import Distribution, as_tensor, common_device, default_device

__all__ = ['Normal', 'Bernoulli', 'Categorical', 'Discrete', 'LinearLogits',
           'linear_logits', 'LinearClassLogits', 'linear_class_logits']

_FLOATS = (torch.float16, torch.float32, torch.float64)
_INTS = (torch.int16, torch.int32, torch.int64)


def _assert_same_float_dtype(tensors_with_name):
    """distributions/utils.py:146-190 (message kept)."""
    dtype = None
    for t, name in tensors_with_name:
        if t.dtype not in _FLOATS:
            raise TypeError("{}({}) must have a float dtype.".format(
                name, t.dtype))
        if dtype is None:
            dtype = t.dtype
        elif t.dtype != dtype:
            raise TypeError(
                "{} must have the same dtype as {}.".format(
                    name, tensors_with_name[0][1]))
    return dtype


_scalar_cache = {}


def _scalar_param(value, device):
    """A Python number given as a distribution parameter -> ONE float32
    device tensor per (value, device), reused by every later construction
    (SHARED, and exposed as `dist.mean` / `.std` / `.logstd`: treat it as
    read-only -- an in-place op on it changes every later distribution built
    from the same number): a
    model function is re-evaluated on every run, and a fresh tensor each time
    would cost a host-to-device copy per evaluation and defeat the native
    plans' "parameters unchanged since last run" check (which goes by
    storage and version)."""
    key = (float(value), str(device))
    t = _scalar_cache.get(key)
    if t is None:
        if len(_scalar_cache) > 256:
            _scalar_cache.clear()
        # (made OUTSIDE inference mode whatever the caller is in: a cached
        # inference tensor would break later autograd use and has no version
        # counter for the native plans' "unchanged" check)
        with torch.inference_mode(False):
            t = _scalar_cache[key] = torch.tensor(float(value),
                                                  dtype=torch.float32,
                                                  device=device)
    return t


def _as_param(value, device):
    if isinstance(value, (int, float)) and not isinstance(value, bool):
        return _scalar_param(value, device)
    return as_tensor(value, dtype=None if isinstance(value, torch.Tensor)
                     else torch.float32, device=device)


def _require_f32(dtype, what):
    if dtype != torch.float32:
        raise TypeError(
            "{}: the MI355X kernels compute in float32 only (HMC is float32 "
            "in the reference, hmc.py:22,72-87); got {}.".format(what, dtype))


class Normal(Distribution):
    """Univariate Normal (univariate.py:43-184)."""

    def __init__(self, mean=0., _sentinel=None, std=None, logstd=None,
                 group_ndims=0, is_reparameterized=True,
                 use_path_derivative=False, check_numerics=False, **kwargs):
        if _sentinel is not None:
            raise ValueError(
                "The order of logstd/std has changed to std/logstd since "
                "0.3.1. Please use named arguments: Normal(mean, std=..., "
                "...) or Normal(mean, logstd=..., ...).")
        if (logstd is None) == (std is None):
            raise ValueError(
                "Either `std` or `logstd` should be passed. It is not allowed "
                "that both are specified or both are not.")
        dev = common_device(mean, std, logstd) or default_device()
        f32 = torch.float32
        # (a mean that is sigmoid(gathered_dot(latent, ...)) stays a symbol:
        # the rating likelihood of pmf_hmc.py:26-31, which the native
        # gathered-dot plan evaluates without materialising it; any use of it
        # as a tensor forces it)
        number = isinstance(mean, (int, float)) and not isinstance(mean, bool)
        self._mean = _scalar_param(mean, dev) if number else as_tensor(
            mean, dtype=None if isinstance(mean, torch.Tensor) else f32,
            device=dev,
            keep_symbolic=_symbolic.gathered_dot_mean(mean) is not None)
        # The parameter that was not given is derived on first use (a model
        # function is re-evaluated on every transition: an eager exp / log
        # would be one more kernel launch per evaluation), unless
        # check_numerics wants it inspected here.
        if logstd is None:
            self._std = _as_param(std, dev)
            dtype = _assert_same_float_dtype([(self._mean, 'Normal.mean'),
                                              (self._std, 'Normal.std')])
            self._logstd = None                          # log(std), :99
            if check_numerics and not bool(
                    torch.isfinite(self.logstd).all()):
                raise FloatingPointError("log(std) : Tensor had Inf or NaN")
            given = self._std
            self._given_spread = ('std', given)
        else:
            self._logstd = _as_param(logstd, dev)
            dtype = _assert_same_float_dtype(
                [(self._mean, 'Normal.mean'),
                 (self._logstd, 'Normal.logstd')])
            self._std = None                             # exp(logstd), :108
            if check_numerics and not bool(torch.isfinite(self.std).all()):
                raise FloatingPointError("exp(logstd) : Tensor had Inf or NaN")
            given = self._logstd
            self._given_spread = ('logstd', given)
        _require_f32(dtype, 'Normal')
        try:
            self._batch_shape_static = broadcast_shapes(self._mean.shape,
                                                        given.shape)
        except RuntimeError:
            raise ValueError(
                "mean and std/logstd should be broadcastable to match each "
                "other. ({} vs. {})".format(tuple(self._mean.shape),
                                            tuple(given.shape)))
        self._check_numerics = check_numerics
        super(Normal, self).__init__(
            dtype=dtype, param_dtype=dtype, is_continuous=True,
            is_reparameterized=is_reparameterized,
            use_path_derivative=use_path_derivative,
            group_ndims=group_ndims, **kwargs)

    @property
    def mean(self):
        return self._mean

    @property
    def given_spread(self):
        """('std' | 'logstd', tensor): the parameter the constructor was
        given (the other one is derived on first use)."""
        return self._given_spread

    # The parameter the constructor was not given is derived on first use
    # and kept -- unless that first use happens with autograd off while the
    # given one requires grad (an evaluation pass under torch.no_grad()): a
    # cached tensor without grad_fn would silently cut every later
    # differentiable use of this distribution from its parameter.
    @property
    def logstd(self):
        if self._logstd is not None:
            return self._logstd
        out = torch.log(self._std)
        if torch.is_grad_enabled() or not self._std.requires_grad:
            self._logstd = out
        return out

    @property
    def std(self):
        if self._std is not None:
            return self._std
        out = torch.exp(self._logstd)
        if torch.is_grad_enabled() or not self._logstd.requires_grad:
            self._std = out
        return out

    def _device(self):
        return self._mean.device

    def _get_value_shape(self):
        return torch.Size([])

    def _get_batch_shape(self):
        return self._batch_shape_static

    def _sample(self, n_samples):
        """univariate.py:161-172 on the Philox STREAM_DIST stream."""
        mean, std = self._mean, self.std
        if not self.is_reparameterized:
            mean, std = mean.detach(), std.detach()
        batch = self._get_batch_shape()
        m = mean.expand(batch).contiguous() if mean.numel() != 1 else mean
        s = std.expand(batch).contiguous() if std.numel() != 1 else std
        _ops.require_device(m, s)
        inner = 1
        for d in batch:
            inner *= int(d)
        n = int(n_samples) * inner
        eps_shape = (int(n_samples),) + tuple(batch)
        seed, offset = next_op_offset()
        if self.is_reparameterized and (mean.requires_grad or
                                        std.requires_grad):
            # reparameterisation trick: draw N(0,1) with the kernel, then
            # let autograd see  eps * std + mean
            one = torch.ones(1, device=m.device)
            zero = torch.zeros(1, device=m.device)
            eps = torch.empty(eps_shape, dtype=torch.float32, device=m.device)
            _capi.call('zshmc_normal_sample', eps.data_ptr(), zero.data_ptr(),
                       one.data_ptr(), n, max(inner, 1), _capi.BCAST_SCALAR,
                       _capi.BCAST_SCALAR, seed, offset,
                       _capi.current_stream())
            return eps * std + mean
        out = torch.empty(eps_shape, dtype=torch.float32, device=m.device)
        _capi.call(
            'zshmc_normal_sample', out.data_ptr(), m.data_ptr(), s.data_ptr(),
            n, max(inner, 1),
            _capi.BCAST_SCALAR if m.numel() == 1 else _capi.BCAST_FULL,
            _capi.BCAST_SCALAR if s.numel() == 1 else _capi.BCAST_FULL,
            seed, offset, _capi.current_stream())
        return out

    def _log_prob_grouped(self, given):
        mean = self.path_param(self._mean)
        logstd = self.path_param(self.logstd)
        full = broadcast_shapes(given.shape, mean.shape, logstd.shape)
        if self._group_ndims > len(full):
            raise ValueError("group_ndims {} exceeds log_prob rank {}"
                             .format(self._group_ndims, len(full)))
        out = _ops.NormalLogProb.apply(given, mean, logstd, self._group_ndims)
        if self._check_numerics and not bool(torch.isfinite(
                torch.exp(-2 * logstd)).all()):
            raise FloatingPointError("precision : Tensor had Inf or NaN")
        return out

    def _log_prob(self, given):
        return _ops.NormalLogProb.apply(
            given, self.path_param(self._mean),
            self.path_param(self.logstd), 0)


class LinearLogits(object):
    """Lazy `w @ X^T` -- or a sum of such terms over several latents plus a
    per-chain bias, `w1 @ X1^T + w2 @ X2^T + b` -- : logits of shape
    w.shape[:-1] + [n_rows] that are never materialised.
    `Bernoulli(linear_logits(w, X), group_ndims=1)` evaluates log_prob and
    its gradient with the fused fp32-MFMA kernels (csrc/linear_bernoulli.hip,
    linear_bernoulli_wide.hip; up to 1024 features in total); anything else
    falls back to `.dense()`.

    `terms`: [(w_k, X_k, scalar_k)]; X_k None stands for a column of ones
    (w_k is a bias: [..., 1], or [...] when scalar_k)."""

    def __init__(self, w, X, bias=None):
        w = as_tensor(w)          # (a symbolic latent: the latent itself)
        X = as_tensor(X)
        if X.dim() != 2 or w.dim() < 1 or w.shape[-1] != X.shape[-1]:
            raise ValueError(
                "linear_logits: w[..., D] and X[N, D] expected, got {} and {}"
                .format(tuple(w.shape), tuple(X.shape)))
        self.terms = [(w, X, False)]
        if bias is not None:
            b = as_tensor(bias)
            lead = tuple(w.shape[:-1])
            if tuple(b.shape) == lead + (1,):
                self.terms.append((b, None, False))
            elif tuple(b.shape) == lead:
                self.terms.append((b, None, True))
            else:
                raise ValueError(
                    "linear_logits: bias of shape {} or {} expected, got {}"
                    .format(lead + (1,), lead, tuple(b.shape)))

    @classmethod
    def of_terms(cls, terms):
        """From [(w_k, X_k | None, scalar_k)] with equal leading shapes (the
        lowering of the literal spelling, _symbolic.lower_bernoulli_logits)."""
        self = cls.__new__(cls)
        self.terms = [(as_tensor(w), None if X is None else as_tensor(X),
                       bool(sc)) for w, X, sc in terms]
        lead = {tuple(w.shape if sc else w.shape[:-1])
                for w, _, sc in self.terms}
        rows = {int(X.shape[0]) for _, X, _ in self.terms if X is not None}
        if len(lead) != 1 or len(rows) != 1 or any(
                X is not None and (X.dim() != 2 or sc or
                                   w.shape[-1] != X.shape[-1])
                for w, X, sc in self.terms):
            raise ValueError("linear_logits: inconsistent terms")
        return self

    # the single-term view (one weight latent, no bias)
    @property
    def w(self):
        return self.terms[0][0]

    @property
    def X(self):
        return self.terms[0][1]

    @property
    def single(self):
        return len(self.terms) == 1

    @property
    def n_rows(self):
        return next(int(X.shape[0]) for _, X, _ in self.terms
                    if X is not None)

    @property
    def n_features(self):
        return sum(1 if X is None else int(X.shape[-1])
                   for _, X, _ in self.terms)

    @property
    def lead_shape(self):
        w, _, sc = self.terms[0]
        return tuple(w.shape if sc else w.shape[:-1])

    @property
    def shape(self):
        return torch.Size(self.lead_shape + (self.n_rows,))

    @property
    def dtype(self):
        return self.w.dtype

    @property
    def device(self):
        return self.w.device

    def design_requires_grad(self):
        return any(X is not None and X.requires_grad for _, X, _ in self.terms)

    def packed(self):
        """(w [..., D_total], X [N, D_total]): the terms side by side -- the
        weights by torch.cat (differentiable), the design matrices cached."""
        if self.single:
            return self.w, self.X
        ws = [w.unsqueeze(-1) if sc else w for w, _, sc in self.terms]
        return torch.cat(ws, -1), _ops.packed_design(
            [X for _, X, _ in self.terms], self.n_rows, self.w.device)

    def dense(self):
        out = None
        for w, X, sc in self.terms:
            t = (w.unsqueeze(-1) if sc else w) if X is None else w @ X.t()
            out = t if out is None else out + t
        return out


def linear_logits(w, X, bias=None):
    return LinearLogits(w, X, bias)


class Bernoulli(Distribution):
    """Univariate Bernoulli (univariate.py:334-406)."""

    def __init__(self, logits, dtype=torch.int32, group_ndims=0, **kwargs):
        self._lazy = None
        logits = _symbolic.lower_bernoulli_logits(logits)
        if isinstance(logits, LinearLogits):
            if logits.dtype != torch.float32:
                raise TypeError("Bernoulli: linear_logits must be float32")
            self._lazy = logits
            if dtype not in _FLOATS + _INTS:
                raise TypeError(
                    "`dtype`({}) must be int or float.".format(dtype))
            self._logits = None
            super(Bernoulli, self).__init__(
                dtype=dtype, param_dtype=torch.float32, is_continuous=False,
                is_reparameterized=False, group_ndims=group_ndims, **kwargs)
            return
        dev = common_device(logits) or default_device()
        self._logits = as_tensor(logits, dtype=None if isinstance(
            logits, torch.Tensor) else torch.float32, device=dev)
        param_dtype = _assert_same_float_dtype(
            [(self._logits, 'Bernoulli.logits')])
        _require_f32(param_dtype, 'Bernoulli')
        if dtype not in _FLOATS + _INTS:
            raise TypeError("`dtype`({}) must be int or float.".format(dtype))
        super(Bernoulli, self).__init__(
            dtype=dtype, param_dtype=param_dtype, is_continuous=False,
            is_reparameterized=False, group_ndims=group_ndims, **kwargs)

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._lazy.dense()
        return self._logits

    def _device(self):
        return self._lazy.device if self._lazy is not None \
            else self._logits.device

    def _get_value_shape(self):
        return torch.Size([])

    def _get_batch_shape(self):
        return self._lazy.shape if self._lazy is not None \
            else self._logits.shape

    def _sample(self, n_samples):
        """univariate.py:386-396: U[0,1) < sigmoid(logits)."""
        logits = self.logits.detach().contiguous()
        _ops.require_device(logits)
        inner = max(logits.numel(), 1)
        n = int(n_samples) * inner
        out = torch.empty((int(n_samples),) + tuple(logits.shape),
                          dtype=torch.int32, device=logits.device)
        seed, offset = next_op_offset()
        _capi.call('zshmc_bernoulli_sample', out.data_ptr(),
                   logits.data_ptr(), n, inner, seed, offset,
                   _capi.current_stream())
        return out if self.dtype == torch.int32 else out.to(self.dtype)

    def _log_prob_grouped(self, given):
        given = given.to(self.param_dtype)          # :399
        lazy = self._lazy
        if (lazy is not None and self._group_ndims >= 1 and
                given.dim() == 1 and given.shape[0] == lazy.n_rows and
                # the fused kernel differentiates w.r.t. w only: a design
                # matrix that needs a gradient takes the dense path
                not lazy.design_requires_grad() and not given.requires_grad and
                lazy.n_features <= _ops.MAX_LIKELIHOOD_WIDTH and
                len(lazy.lead_shape) >= self._group_ndims - 1):
            w_all, x_all = lazy.packed()
            ll = _ops.LinearBernoulliLogLik.apply(w_all, x_all, given)
            extra = self._group_ndims - 1
            return ll if extra == 0 else ll.sum(
                dim=tuple(range(-extra, 0)))
        if self._logits is None:
            self._logits = lazy.dense()
        try:
            full = broadcast_shapes(given.shape, self._logits.shape)
        except RuntimeError:
            raise ValueError(
                "given and logits cannot broadcast to match. ({} vs. {})"
                .format(tuple(given.shape), tuple(self._logits.shape)))
        if self._group_ndims > len(full):
            raise ValueError("group_ndims {} exceeds log_prob rank {}"
                             .format(self._group_ndims, len(full)))
        return _ops.BernoulliLogProb.apply(self._logits, given,
                                           self._group_ndims)

    def _log_prob(self, given):
        return _ops.BernoulliLogProb.apply(
            self.logits, given.to(self.param_dtype), 0)


class LinearClassLogits(object):
    """Lazy class logits of a softmax regression,
    `logits[..., n, k] = sum_f X[n, f] * w[..., k, f]` -- what the reference
    spells `tf.matmul(X, w, transpose_b=True)` on a [chains, N, F] tiling of X
    -- of shape w.shape[:-2] + [N, K], never materialised:
    `Categorical(linear_class_logits(w, X), group_ndims=1)` evaluates log_prob
    and its gradient with the fused fp32-MFMA kernels
    (zshmc_linear_categorical_log_lik; up to 32 classes x 1024 features);
    anything else falls back to `.dense()`."""

    def __init__(self, w, X):
        w = as_tensor(w)          # (a symbolic latent: the latent itself)
        X = as_tensor(X)
        if X.dim() != 2 or w.dim() < 2 or w.shape[-1] != X.shape[-1]:
            raise ValueError(
                "linear_class_logits: w[..., K, F] and X[N, F] expected, got "
                "{} and {}".format(tuple(w.shape), tuple(X.shape)))
        self.w, self.X = w, X

    @property
    def n_rows(self):
        return int(self.X.shape[0])

    @property
    def n_classes(self):
        return int(self.w.shape[-2])

    @property
    def n_features(self):
        return int(self.w.shape[-1])

    @property
    def lead_shape(self):
        return tuple(self.w.shape[:-2])

    @property
    def shape(self):
        return torch.Size(self.lead_shape + (self.n_rows, self.n_classes))

    @property
    def dtype(self):
        return self.w.dtype

    @property
    def device(self):
        return self.w.device

    def fused_ok(self):
        return (self.n_classes <= _ops.MAX_CLASSES and
                self.n_features <= _ops.MAX_LIKELIHOOD_WIDTH and
                not self.X.requires_grad and self.dtype == torch.float32)

    def dense(self):
        return torch.matmul(self.X, self.w.transpose(-1, -2))


def linear_class_logits(w, X):
    return LinearClassLogits(w, X)


class Categorical(Distribution):
    """Univariate Categorical (univariate.py:409-551)."""

    def __init__(self, logits, dtype=torch.int32, group_ndims=0, **kwargs):
        self._lazy = None
        logits = _symbolic.lower_categorical_logits(logits)
        if isinstance(logits, LinearClassLogits):
            if logits.dtype != torch.float32:
                raise TypeError(
                    "Categorical: linear_class_logits must be float32")
            if dtype not in (torch.float32, torch.float64, torch.int32,
                             torch.int64):
                raise TypeError(
                    "`dtype`({}) not in allowed dtypes.".format(dtype))
            self._lazy = logits
            self._logits_dense = None
            self._n_categories = logits.n_classes
            super(Categorical, self).__init__(
                dtype=dtype, param_dtype=torch.float32, is_continuous=False,
                is_reparameterized=False, group_ndims=group_ndims, **kwargs)
            return
        dev = common_device(logits) or default_device()
        self._logits = as_tensor(logits, dtype=None if isinstance(
            logits, torch.Tensor) else torch.float32, device=dev)
        if self._logits.dtype not in (torch.float32, torch.float64):
            raise TypeError(
                "Categorical.logits({}) must be float32 or float64.".format(
                    self._logits.dtype))
        _require_f32(self._logits.dtype, 'Categorical')
        if dtype not in (torch.float32, torch.float64, torch.int32,
                         torch.int64):
            raise TypeError(
                "`dtype`({}) not in allowed dtypes.".format(dtype))
        if self._logits.dim() < 1:
            raise ValueError(
                "Categorical.logits should have rank >= 1, got a scalar.")
        self._n_categories = int(self._logits.shape[-1])
        self._logits_dense = self._logits
        super(Categorical, self).__init__(
            dtype=dtype, param_dtype=self._logits.dtype, is_continuous=False,
            is_reparameterized=False, group_ndims=group_ndims, **kwargs)

    @property
    def _logits(self):
        if self._logits_dense is None:
            self._logits_dense = self._lazy.dense()
        return self._logits_dense

    @_logits.setter
    def _logits(self, value):
        self._logits_dense = value

    @property
    def logits(self):
        return self._logits

    @property
    def n_categories(self):
        return self._n_categories

    def _device(self):
        return self._lazy.device if self._lazy is not None \
            else self._logits.device

    def _get_value_shape(self):
        return torch.Size([])

    def _get_batch_shape(self):
        return self._lazy.shape[:-1] if self._lazy is not None \
            else self._logits.shape[:-1]

    def _log_prob_grouped(self, given):
        lazy = self._lazy
        if (lazy is not None and self._group_ndims >= 1 and lazy.fused_ok()
                and not given.requires_grad
                and given.numel() == lazy.n_rows and given.dim() >= 1
                and given.shape[-1] == lazy.n_rows
                and len(lazy.lead_shape) >= self._group_ndims - 1):
            # labels [N] (or [1, ..., N]) shared by every chain, the data rows
            # the innermost grouped axis: the fused likelihood
            labels = _ops.labels_as_float(given, lazy.n_classes)
            ll = _ops.LinearCategoricalLogLik.apply(lazy.w, lazy.X, labels)
            extra = self._group_ndims - 1
            return ll if extra == 0 else ll.sum(
                dim=tuple(range(-extra, 0)))
        return super(Categorical, self)._log_prob_grouped(given)

    def _sample(self, n_samples):
        """univariate.py:478-494; inverse-CDF on the Philox stream."""
        logits = self._logits.detach().contiguous()
        _ops.require_device(logits)
        batch = tuple(logits.shape[:-1])
        rows = 1
        for d in batch:
            rows *= int(d)
        out = torch.empty((int(n_samples),) + batch, dtype=torch.int32,
                          device=logits.device)
        seed, offset = next_op_offset()
        _capi.call('zshmc_categorical_sample', out.data_ptr(),
                   logits.data_ptr(), int(n_samples), rows,
                   self._n_categories, seed, offset, _capi.current_stream())
        return out if self.dtype == torch.int32 else out.to(self.dtype)

    def _log_prob(self, given):
        # :499-505 explicit broadcast of given vs logits[..., :-1]
        try:
            batch = broadcast_shapes(given.shape,
                                           self._logits.shape[:-1])
        except RuntimeError:
            raise ValueError(
                "given and logits cannot broadcast to match. ({} vs. {})"
                .format(tuple(given.shape), tuple(self._logits.shape)))
        labels = given.expand(batch).to(torch.int64).contiguous().reshape(-1)
        logits = self._logits.expand(tuple(batch) + (self._n_categories,)) \
            .contiguous().reshape(-1, self._n_categories)
        out = _ops.CategoricalLogProb.apply(logits, labels)
        return out.reshape(batch)


Discrete = Categorical
