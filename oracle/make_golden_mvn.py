#!/usr/bin/env python
"""Generate tests/golden/mvn_vectors.npz.  Run in the build container:

    python oracle/make_golden_mvn.py

The parameter recipe and the expected values are the reference's own test
(tests/distributions/test_multivariate.py:54-64 `_gen_test_params`, seeds 23 /
233 / 2333 of :93,:119; expected log-density = scipy.stats.multivariate_normal
.logpdf exactly as :112-115), in float64.  The reference draws the evaluation
points with its own TF sampler (:104, not reproducible here); they are drawn
with NumPy from the same distribution instead -- any point pins the density."""
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, '..', 'tests', 'golden')


def gen_test_params(seed):
    np.random.seed(seed)
    mean = 10 * np.random.normal(size=(10, 11, 3)).astype('d')
    cov = np.zeros((10, 11, 3, 3))
    cov_chol = np.zeros_like(cov)
    for i in range(10):
        for j in range(11):
            cov[i, j] = stats.invwishart.rvs(3, np.eye(3))
            cov[i, j] /= np.max(np.diag(cov[i, j]))
            cov_chol[i, j, :, :] = np.linalg.cholesky(cov[i, j])
    return mean, cov, cov_chol


def main():
    out = {}
    n_exp = 12
    for seed in (23, 233, 2333):
        mean, cov, chol = gen_test_params(seed)
        rng = np.random.RandomState(seed + 1)
        noise = rng.normal(size=(n_exp, 10, 11, 3))
        samples = mean + np.einsum('bcij,nbcj->nbci', chol, noise)
        logpdf = np.zeros((n_exp, 10, 11))
        for i in range(10):
            for j in range(11):
                logpdf[:, i, j] = stats.multivariate_normal.logpdf(
                    samples[:, i, j, :], mean[i, j], cov[i, j])
        k = 's%d_' % seed
        out[k + 'mean'], out[k + 'cov'], out[k + 'chol'] = mean, cov, chol
        out[k + 'samples'], out[k + 'logpdf'] = samples, logpdf
    # one larger, shared factor (the HMC use: many chains, one covariance)
    rng = np.random.RandomState(7)
    D = 24
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + 0.3 * np.eye(D)
    chol = np.linalg.cholesky(cov)
    mean = rng.normal(size=D)
    x = mean + rng.normal(size=(50, D)) @ chol.T
    out['big_mean'], out['big_chol'], out['big_cov'] = mean, chol, cov
    out['big_x'] = x
    out['big_logpdf'] = stats.multivariate_normal.logpdf(x, mean, cov)
    out['big_grad'] = -np.linalg.solve(cov, (x - mean).T).T
    np.savez_compressed(os.path.join(GOLD, 'mvn_vectors.npz'), **out)
    print('wrote mvn_vectors.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
