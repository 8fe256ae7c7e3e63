"""The MultivariateNormalCholesky oracle (oracle/distributions_ref.py) against
the reference's own test vectors (tests/distributions/test_multivariate.py:
54-64,96-120: invwishart covariances, scipy logpdf), tests/golden/
mvn_vectors.npz."""
import os

import numpy as np

from oracle import distributions_ref as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'mvn_vectors.npz')


def test_log_prob_matches_reference_test_vectors():
    g = np.load(GOLD)
    for seed in (23, 233, 2333):
        k = 's%d_' % seed
        # float64, as the reference's test feeds it: its assertAllClose
        # default rtol = atol = 1e-6
        d64 = R.MultivariateNormalCholesky(g[k + 'mean'], g[k + 'chol'],
                                           dtype=np.float64)
        np.testing.assert_allclose(d64.log_prob(g[k + 'samples']),
                                   g[k + 'logpdf'], rtol=1e-9, atol=1e-9)
        # float32 (the device's arithmetic): means ~ 10 and invwishart(3)
        # factors as ill-conditioned as 1e3 leave ~1e-3 relative
        d = R.MultivariateNormalCholesky(g[k + 'mean'], g[k + 'chol'])
        got = d.log_prob(g[k + 'samples'])
        np.testing.assert_allclose(got, g[k + 'logpdf'], rtol=2e-3, atol=2e-3)
        assert got.shape == (12, 10, 11) and got.dtype == np.float32


def test_shared_factor_log_prob_and_gradient():
    g = np.load(GOLD)
    d = R.MultivariateNormalCholesky(g['big_mean'], g['big_chol'])
    np.testing.assert_allclose(d.log_prob(g['big_x']), g['big_logpdf'],
                               rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(d.grad_given(g['big_x']), g['big_grad'],
                               rtol=2e-3, atol=2e-3)
    # group_ndims sums batch axes (base.py:302-304)
    k = 's23_'
    d1 = R.MultivariateNormalCholesky(g[k + 'mean'], g[k + 'chol'],
                                      group_ndims=1)
    np.testing.assert_allclose(d1.log_prob(g[k + 'samples']),
                               g[k + 'logpdf'].sum(-1), rtol=2e-3, atol=2e-2)
    d64 = R.MultivariateNormalCholesky(g['big_mean'], g['big_chol'],
                                       dtype=np.float64)
    np.testing.assert_allclose(d64.log_prob(g['big_x']), g['big_logpdf'],
                               rtol=1e-10)
    np.testing.assert_allclose(d64.grad_given(g['big_x']), g['big_grad'],
                               rtol=1e-8, atol=1e-9)


def test_sample_moments():
    g = np.load(GOLD)
    d = R.MultivariateNormalCholesky(g['s23_mean'][:2, :3], g['s23_chol'][:2, :3])
    s = d.sample(20000, seed=5, offset=3)
    assert s.shape == (20000, 2, 3, 3)
    np.testing.assert_allclose(s.mean(0), g['s23_mean'][:2, :3], rtol=5e-2,
                               atol=5e-2)
    for i in range(2):
        for j in range(3):
            np.testing.assert_allclose(np.cov(s[:, i, j].T),
                                       g['s23_cov'][i, j], rtol=1e-1, atol=1e-1)
    assert d.sample(None, seed=1).shape == (2, 3, 3)
