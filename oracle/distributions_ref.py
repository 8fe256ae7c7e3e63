"""NumPy float32 restatement of the reference's Normal / Bernoulli /
Categorical / UnnormalizedMultinomial log_prob (+ analytic gradients, +
sampling on the shared Philox stream).  TEST INFRASTRUCTURE (see
oracle/__init__.py).

Follows, in /root/reference:
  zhusuan/distributions/base.py:236-263   Distribution.sample (squeeze rule)
  zhusuan/distributions/base.py:290-304   Distribution.log_prob (group_ndims sum)
  zhusuan/distributions/univariate.py:96-103,174-181   Normal
  zhusuan/distributions/univariate.py:386-403          Bernoulli
  zhusuan/distributions/univariate.py:478-548          Categorical
  zhusuan/distributions/multivariate.py:435-443        UnnormalizedMultinomial

The closed forms that the reference delegates to TensorFlow
(tf.nn.sigmoid_cross_entropy_with_logits, sparse_softmax_cross_entropy_with_logits,
tf.reduce_logsumexp; TensorFlow>=1.13, un-vendored) are restated from their
documented formulas:  sce(l, z) = max(l,0) - l*z + log1p(exp(-|l|));
ssce(l, k) = logsumexp(l) - l[k].
"""
import numpy as np

from . import philox

F32 = np.float32
LOG_2PI_HALF = F32(-0.5 * np.log(2 * np.pi))


def _group_sum(log_p, group_ndims):
    # base.py:302-304 -- reduce_sum over the last `group_ndims` axes.
    if group_ndims == 0:
        return log_p
    axes = tuple(range(-group_ndims, 0))
    return np.sum(log_p, axis=axes, dtype=log_p.dtype)


class Normal(object):
    """univariate.py:43-184."""

    def __init__(self, mean=0., std=None, logstd=None, group_ndims=0):
        if (logstd is None) == (std is None):
            raise ValueError(
                "Either `std` or `logstd` should be passed. It is not allowed "
                "that both are specified or both are not.")
        self.mean = np.asarray(mean, dtype=F32)
        if logstd is None:
            self.std = np.asarray(std, dtype=F32)
            self.logstd = np.log(self.std)            # :99
        else:
            self.logstd = np.asarray(logstd, dtype=F32)
            self.std = np.exp(self.logstd)            # :108
        self.group_ndims = group_ndims

    def _log_prob(self, given):
        # :174-181
        precision = np.exp(F32(-2) * self.logstd)
        return (LOG_2PI_HALF - self.logstd -
                F32(0.5) * precision * np.square(given - self.mean))

    def log_prob(self, given):
        given = np.asarray(given, dtype=F32)
        return _group_sum(self._log_prob(given), self.group_ndims)

    def grad_given(self, given):
        """d log_prob / d given, elementwise (what tf.gradients yields):
        -precision * (given - mean)."""
        given = np.asarray(given, dtype=F32)
        precision = np.exp(F32(-2) * self.logstd)
        return -(precision * (given - self.mean))

    def grad_params(self, given):
        """(d/dmean, d/dlogstd) elementwise, before broadcasting reduction."""
        given = np.asarray(given, dtype=F32)
        precision = np.exp(F32(-2) * self.logstd)
        d = given - self.mean
        return precision * d, F32(-1) + precision * d * d

    def sample(self, n_samples=None, seed=0, offset=0):
        # :161-172 ; base.py:251-256 (None -> squeeze axis 0)
        n = 1 if n_samples is None else int(n_samples)
        batch = np.broadcast(self.mean, self.std).shape
        shape = (n,) + tuple(batch)
        z = philox.normal_flat(seed, offset, int(np.prod(shape))).reshape(shape)
        s = z * self.std + self.mean
        return s[0] if n_samples is None else s


def _sigmoid(l):
    with np.errstate(over='ignore'):      # exp(-l) -> inf gives the limit 0
        return F32(1) / (F32(1) + np.exp(-l))


class Bernoulli(object):
    """univariate.py:334-406."""

    def __init__(self, logits, dtype=np.int32, group_ndims=0):
        self.logits = np.asarray(logits, dtype=F32)
        self.dtype = dtype
        self.group_ndims = group_ndims

    def _log_prob(self, given):
        # :398-403 ; -sigmoid_cross_entropy_with_logits(labels=z, logits=l)
        z = np.asarray(given).astype(F32)
        z, l = np.broadcast_arrays(z, self.logits)
        return -(np.maximum(l, F32(0)) - l * z +
                 np.log1p(np.exp(-np.abs(l))))

    def log_prob(self, given):
        return _group_sum(self._log_prob(given), self.group_ndims)

    def grad_logits(self, given):
        z = np.asarray(given).astype(F32)
        z, l = np.broadcast_arrays(z, self.logits)
        return z - _sigmoid(l)

    def sample(self, n_samples=None, seed=0, offset=0):
        # :386-396   alpha ~ U[0,1);  sample = alpha < sigmoid(logits)
        n = 1 if n_samples is None else int(n_samples)
        shape = (n,) + self.logits.shape
        alpha = philox.uniform_flat(seed, offset,
                                    int(np.prod(shape))).reshape(shape)
        s = (alpha < _sigmoid(self.logits)).astype(self.dtype)
        return s[0] if n_samples is None else s


def _logsumexp(l, axis=-1, keepdims=False):
    m = np.max(l, axis=axis, keepdims=True)
    out = m + np.log(np.sum(np.exp(l - m), axis=axis, keepdims=True,
                            dtype=l.dtype))
    return out if keepdims else np.squeeze(out, axis=axis)


class Categorical(object):
    """univariate.py:409-551."""

    def __init__(self, logits, dtype=np.int32, group_ndims=0):
        self.logits = np.asarray(logits, dtype=F32)
        if self.logits.ndim < 1:
            raise ValueError("Categorical.logits should be at least 1-D")
        self.n_categories = self.logits.shape[-1]
        self.dtype = dtype
        self.group_ndims = group_ndims

    def _broadcast(self, given):
        # :499-505 -- given * ones(batch), logits * ones(given[..., None])
        given = np.asarray(given)
        shape = np.broadcast(np.empty(given.shape),
                             np.empty(self.logits.shape[:-1])).shape
        given = np.broadcast_to(given, shape).astype(np.int64)
        logits = np.broadcast_to(self.logits, shape + (self.n_categories,))
        return given, logits

    def _log_prob(self, given):
        # :496-548 ; -sparse_softmax_cross_entropy_with_logits
        given, logits = self._broadcast(given)
        picked = np.take_along_axis(logits, given[..., None], axis=-1)[..., 0]
        return picked - _logsumexp(logits, axis=-1)

    def log_prob(self, given):
        return _group_sum(self._log_prob(given), self.group_ndims)

    def grad_logits(self, given):
        given, logits = self._broadcast(given)
        soft = np.exp(logits - _logsumexp(logits, axis=-1, keepdims=True))
        onehot = np.zeros_like(soft)
        np.put_along_axis(onehot, given[..., None], F32(1), axis=-1)
        return onehot - soft

    def sample(self, n_samples=None, seed=0, offset=0):
        """Inverse-CDF sampling on the shared Philox stream (the reference
        calls tf.random.categorical, :483-484, whose stream is unpinned):
        k = #{j : cdf_j <= u}, cdf from softmax in float32, clipped to
        n_categories-1.  Shape [n] + batch (:478-494)."""
        n = 1 if n_samples is None else int(n_samples)
        batch = self.logits.shape[:-1]
        shape = (n,) + batch
        u = philox.uniform_flat(seed, offset, int(np.prod(shape))).reshape(shape)
        soft = np.exp(self.logits - _logsumexp(self.logits, -1, keepdims=True))
        cdf = np.cumsum(soft, axis=-1, dtype=F32)
        k = np.sum(cdf[None] <= u[..., None], axis=-1)
        k = np.minimum(k, self.n_categories - 1).astype(self.dtype)
        return k[0] if n_samples is None else k


class UnnormalizedMultinomial(object):
    """multivariate.py:339-446."""

    def __init__(self, logits, normalize_logits=True, dtype=np.int32,
                 group_ndims=0):
        self.logits = np.asarray(logits, dtype=F32)
        self.normalize_logits = normalize_logits
        self.group_ndims = group_ndims

    def _log_prob(self, given):
        # :435-443
        given = np.asarray(given).astype(F32)
        given, logits = np.broadcast_arrays(given, self.logits)
        if self.normalize_logits:
            logits = logits - _logsumexp(logits, axis=-1, keepdims=True)
        return np.sum(given * logits, axis=-1, dtype=F32)

    def log_prob(self, given):
        return _group_sum(self._log_prob(given), self.group_ndims)

    def grad_logits(self, given):
        given = np.asarray(given).astype(F32)
        given, logits = np.broadcast_arrays(given, self.logits)
        if not self.normalize_logits:
            return given.copy()
        soft = np.exp(logits - _logsumexp(logits, axis=-1, keepdims=True))
        return given - np.sum(given, axis=-1, keepdims=True) * soft


# ---------------------------------------------------------------------------
# Two-parameter continuous families.  lgamma /
# digamma come from scipy.special (float64, rounded once): the closed forms
# are the reference's, line for line.
# ---------------------------------------------------------------------------
class _TwoParam(object):
    def __init__(self, a, b, group_ndims=0):
        self.a = np.asarray(a, dtype=F32)
        self.b = np.asarray(b, dtype=F32)
        self.group_ndims = group_ndims

    def log_prob(self, given):
        return _group_sum(self._log_prob(np.asarray(given, dtype=F32)),
                          self.group_ndims)


class Laplace(_TwoParam):
    """univariate.py:1164-1277; _log_prob :1268-1275."""

    def _log_prob(self, given):
        return (-np.log(2.) - np.log(self.b.astype(np.float64))
                - np.abs(given.astype(np.float64) - self.a) / self.b).astype(F32)

    def grads(self, given):
        d = np.asarray(given, np.float64) - self.a
        s = np.sign(d)
        b = self.b.astype(np.float64)
        return (-s / b), (s / b), (-1 / b + np.abs(d) / b ** 2)


class Gamma(_TwoParam):
    """univariate.py:662-751; _log_prob :735-748 (rate parameter beta)."""

    def _log_prob(self, given):
        from scipy.special import gammaln
        a, b, x = (v.astype(np.float64) for v in (self.a, self.b, given))
        return (a * np.log(b) - gammaln(a) + (a - 1) * np.log(x) - b * x).astype(F32)

    def grads(self, given):
        from scipy.special import digamma
        a, b, x = (np.asarray(v, np.float64) for v in (self.a, self.b, given))
        return ((a - 1) / x - b, np.log(b) - digamma(a) + np.log(x), a / b - x)


class InverseGamma(_TwoParam):
    """univariate.py:1070-1161; _log_prob :1145-1157."""

    def _log_prob(self, given):
        from scipy.special import gammaln
        a, b, x = (v.astype(np.float64) for v in (self.a, self.b, given))
        return (a * np.log(b) - gammaln(a) - (a + 1) * np.log(x) - b / x).astype(F32)

    def grads(self, given):
        from scipy.special import digamma
        a, b, x = (np.asarray(v, np.float64) for v in (self.a, self.b, given))
        return (-(a + 1) / x + b / x ** 2, np.log(b) - digamma(a) - np.log(x),
                a / b - 1 / x)


class Beta(_TwoParam):
    """univariate.py:753-855; _log_prob :834-853."""

    def _log_prob(self, given):
        from scipy.special import gammaln
        a, b, x = (v.astype(np.float64) for v in (self.a, self.b, given))
        return ((a - 1) * np.log(x) + (b - 1) * np.log1p(-x)
                - (gammaln(a) + gammaln(b) - gammaln(a + b))).astype(F32)

    def grads(self, given):
        from scipy.special import digamma
        a, b, x = (np.asarray(v, np.float64) for v in (self.a, self.b, given))
        return ((a - 1) / x - (b - 1) / (1 - x),
                np.log(x) - digamma(a) + digamma(a + b),
                np.log1p(-x) - digamma(b) + digamma(a + b))


class MultivariateNormalCholesky(object):
    """zhusuan/distributions/multivariate.py:41-193.  float32 restatement:
    forward substitution for matrix_triangular_solve (:183), log_z as :169-174,
    sample = L . noise + mean (:155-157) on the shared Philox stream."""

    def __init__(self, mean, cov_tril, group_ndims=0, dtype=np.float32):
        """dtype=np.float64 evaluates the same restatement in double (the
        reference's own test feeds float64 parameters, test_multivariate.py:
        56), which is how the algorithm is pinned to the scipy vectors; the
        device path is float32."""
        self.dt = np.dtype(dtype).type
        self.mean = np.asarray(mean, self.dt)
        self.cov_tril = np.asarray(cov_tril, self.dt)
        self.n_dim = self.mean.shape[-1]
        assert self.cov_tril.shape == self.mean.shape + (self.n_dim,)
        self.group_ndims = group_ndims

    def _solve_lower(self, L, y):
        """Row-by-row forward substitution, batched over leading axes."""
        D = y.shape[-1]
        z = np.zeros(np.broadcast_shapes(L.shape[:-2], y.shape[:-1]) + (D,),
                     self.dt)
        for i in range(D):
            acc = y[..., i] - (L[..., i, :i] * z[..., :i]).sum(-1,
                                                              dtype=self.dt)
            z[..., i] = acc / L[..., i, i]
        return z

    def _solve_upper_t(self, L, z):
        """w = L^-T z."""
        D = z.shape[-1]
        w = np.zeros_like(z)
        for i in range(D - 1, -1, -1):
            acc = z[..., i] - (L[..., i + 1:, i] * w[..., i + 1:]).sum(
                -1, dtype=self.dt)
            w[..., i] = acc / L[..., i, i]
        return w

    def _log_prob(self, given):
        given = np.asarray(given, self.dt)
        L = self.cov_tril
        log_det = 2 * np.log(np.diagonal(L, axis1=-2, axis2=-1)).sum(
            -1, dtype=self.dt)                                   # :169-170
        log_z = (-self.dt(self.n_dim) / 2 * np.log(
            self.dt(2 * np.pi)) - log_det / 2).astype(self.dt)  # :172-173
        z = self._solve_lower(L, given - self.mean)                 # :180-184
        stoc = -0.5 * np.square(z).sum(-1, dtype=self.dt)        # :185
        return (log_z + stoc).astype(self.dt)

    def log_prob(self, given):
        return _group_sum(self._log_prob(given), self.group_ndims)

    def grad_given(self, given):
        given = np.asarray(given, self.dt)
        z = self._solve_lower(self.cov_tril, given - self.mean)
        return (-self._solve_upper_t(self.cov_tril, z)).astype(self.dt)

    def sample(self, n_samples=None, seed=0, offset=0):
        n = 1 if n_samples is None else int(n_samples)
        shape = (n,) + self.mean.shape
        noise = philox.normal_flat(seed, offset, int(np.prod(shape))).reshape(
            shape)
        out = np.zeros(shape, self.dt)
        for i in range(self.n_dim):
            out[..., i] = (self.cov_tril[..., i, :i + 1] *
                           noise[..., :i + 1]).sum(-1, dtype=self.dt)
        out = out + self.mean
        return out[0] if n_samples is None else out
