"""Pin of the oracle's Laplace / Gamma / InverseGamma / Beta closed forms to
the reference's own test vectors and scipy targets
(tests/distributions/test_univariate.py:690-707, :766-782, :1008-1024,
:1125-1141 of the reference)."""
import numpy as np
import pytest
from scipy import stats

from oracle import distributions_ref as dref

# (class, scipy target, vectors (a, b, given)) exactly as the reference's tests
VECTORS = {
    'Gamma': (lambda g, a, b: stats.gamma.logpdf(g, a, scale=1. / b), [
        (1., 1., [1., 10., 1e8]),
        ([0.5, 1., 2., 3., 5., 7.5, 9.], [2., 2., 2., 1., 0.5, 1., 1.],
         np.transpose([np.arange(1, 20)])),
        ([1e-8, 1e8], [[1., 1e8], [1e-8, 5.]], [7.])]),
    'Beta': (lambda g, a, b: stats.beta.logpdf(g, a, b), [
        ([0.5, 5., 1., 2., 2.], [0.5, 1., 3., 2., 5.],
         np.transpose([np.arange(0.1, 1, 0.1)])),
        ([[1e-8], [1e8]], [[1., 1e8], [1e-8, 1.]], [0.7])]),
    'InverseGamma': (lambda g, a, b: stats.invgamma.logpdf(g, a, scale=b), [
        (1., 1., [1., 10., 1e8]),
        ([0.5, 1., 2., 3., 5., 7.5, 9.], [2., 2., 2., 1., 0.5, 1., 1.],
         np.transpose([np.arange(1, 20)])),
        ([1e-8, 1e8], [[1., 1e8], [1e-8, 5.]], [7.])]),
    'Laplace': (lambda g, a, b: stats.laplace.logpdf(g, a, scale=b), [
        (0., 1., [.01, .1, 1., 10., 100.]),
        ([-3, -2, -1, 0, 1, 2, 3], [.1, 3, 2, 3, 3, 2, .1],
         np.transpose([np.arange(1, 20)])),
        ([1e-5, -1e-5], [[1., 10.], [1e8, 5.]], [7.])]),
}


@pytest.mark.parametrize('name', sorted(VECTORS))
def test_log_prob_matches_reference_vectors(name):
    target, vecs = VECTORS[name]
    for a, b, given in vecs:
        a32, b32, g32 = (np.array(v, np.float32) for v in (a, b, given))
        got = getattr(dref, name)(a32, b32).log_prob(g32)
        want = target(g32, a32, b32)
        # assertAllClose defaults of the reference's tf.test.TestCase
        np.testing.assert_allclose(got, want, rtol=1e-6 * 10, atol=1e-6 * 10 *
                                   max(1.0, np.abs(want).max() * 1e-1))


@pytest.mark.parametrize('name', sorted(VECTORS))
def test_gradients_by_finite_differences(name):
    rng = np.random.RandomState(0)
    a = rng.uniform(0.6, 4.0, size=(5,))
    b = rng.uniform(0.6, 3.0, size=(5,))
    x = rng.uniform(0.1, 0.9, size=(5,)) if name == 'Beta' else \
        rng.uniform(0.3, 4.0, size=(5,))
    d = getattr(dref, name)(a, b)
    gx, ga, gb = d.grads(x)

    def f(a_, b_, x_):
        t, _ = VECTORS[name]
        return t(x_, a_, b_)
    h = 1e-5
    np.testing.assert_allclose(gx, (f(a, b, x + h) - f(a, b, x - h)) / (2 * h),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ga, (f(a + h, b, x) - f(a - h, b, x)) / (2 * h),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gb, (f(a, b + h, x) - f(a, b - h, x)) / (2 * h),
                               rtol=1e-5, atol=1e-6)


def test_group_ndims_sum():
    d = dref.Gamma(np.ones((2, 3), np.float32) * 2, np.ones(3, np.float32),
                   group_ndims=1)
    x = np.arange(1, 7, dtype=np.float32).reshape(2, 3)
    np.testing.assert_allclose(
        d.log_prob(x), stats.gamma.logpdf(x, 2.0).sum(-1), rtol=1e-6)
