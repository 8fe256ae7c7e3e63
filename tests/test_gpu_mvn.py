"""GPU parity of MultivariateNormalCholesky (zhusuan_amd/distributions/
multivariate.py over csrc/mvn.hip) against the oracle restatement of
zhusuan/distributions/multivariate.py:41-193, the reference's own test
vectors (tests/golden/mvn_vectors.npz) and the constructor contracts of
tests/distributions/test_multivariate.py:19-52."""
import os

import numpy as np
import pytest

from oracle import distributions_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'mvn_vectors.npz')


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _T(torch, dev, a):
    return torch.tensor(np.asarray(a, np.float32), device=dev)


def test_init_check_shape_and_inference(env):
    zs, torch, dev = env
    M = zs.distributions.MultivariateNormalCholesky
    with pytest.raises(ValueError, match='should have rank'):
        M(torch.zeros([], device=dev), torch.zeros([], device=dev))
    with pytest.raises(ValueError, match='should have rank'):
        M(torch.zeros([1], device=dev), torch.zeros([1], device=dev))
    with pytest.raises(ValueError, match='compatible'):
        M(torch.zeros([1, 2], device=dev), torch.zeros([1, 2, 3], device=dev))
    d = M(torch.zeros(10, 11, 2, device=dev), torch.zeros(10, 11, 2, 2,
                                                          device=dev))
    assert list(d.get_batch_shape()) == [10, 11]
    assert list(d.get_value_shape()) == [2]
    d = M(torch.ones(2, device=dev), torch.eye(2, device=dev))
    assert list(d.batch_shape) == [] and list(d.value_shape) == [2]


def test_log_prob_reference_vectors(env):
    """test_multivariate.py:96-120 (float32 here)."""
    zs, torch, dev = env
    g = np.load(GOLD)
    for seed in (23, 233, 2333):
        k = 's%d_' % seed
        d = zs.distributions.MultivariateNormalCholesky(
            _T(torch, dev, g[k + 'mean']), _T(torch, dev, g[k + 'chol']),
            check_numerics=True)
        x = _T(torch, dev, g[k + 'samples'])
        lp = d.log_prob(x)
        assert tuple(lp.shape) == (12, 10, 11)
        want32 = R.MultivariateNormalCholesky(g[k + 'mean'],
                                              g[k + 'chol']).log_prob(
            g[k + 'samples'].astype(np.float32))
        # same float32 inputs, different summation order of the solve
        np.testing.assert_allclose(lp.cpu().numpy(), want32, rtol=3e-4,
                                   atol=3e-3)
        np.testing.assert_allclose(lp.cpu().numpy(), g[k + 'logpdf'],
                                   rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(d.prob(x).cpu().numpy(),
                                   np.exp(lp.cpu().numpy()), rtol=1e-5)
        # group_ndims folds batch axes
        d1 = zs.distributions.MultivariateNormalCholesky(
            _T(torch, dev, g[k + 'mean']), _T(torch, dev, g[k + 'chol']),
            group_ndims=2)
        np.testing.assert_allclose(d1.log_prob(x).cpu().numpy(),
                                   g[k + 'logpdf'].sum((-1, -2)), rtol=2e-3)


@pytest.mark.parametrize('D,rows', [(1, 5), (3, 64), (24, 50), (65, 130),
                                    (200, 77), (512, 70)])
def test_shared_factor_log_prob_and_grad(env, D, rows):
    zs, torch, dev = env
    rng = np.random.RandomState(D)
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + 0.5 * np.eye(D)
    chol = np.linalg.cholesky(cov)
    mean = rng.normal(size=D)
    x = mean + rng.normal(size=(rows, D)) @ chol.T
    from scipy import stats
    want = stats.multivariate_normal.logpdf(x, mean, cov).reshape(rows)
    want_g = -np.linalg.solve(cov, (x - mean).T).T
    xt = _T(torch, dev, x).requires_grad_(True)
    d = zs.distributions.MultivariateNormalCholesky(
        _T(torch, dev, mean), _T(torch, dev, chol))
    lp = d.log_prob(xt)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), want, rtol=2e-4,
                               atol=2e-3)
    w = _T(torch, dev, rng.normal(size=rows))
    (lp * w).sum().backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(),
                               want_g * w.cpu().numpy()[:, None], rtol=5e-3,
                               atol=5e-3 * np.abs(want_g).max())
    o = R.MultivariateNormalCholesky(mean, chol)
    np.testing.assert_allclose(lp.detach().cpu().numpy(),
                               o.log_prob(x.astype(np.float32)), rtol=1e-4,
                               atol=1e-3)


def test_parameter_gradients_match_autograd(env):
    zs, torch, dev = env
    rng = np.random.RandomState(3)
    B, D, n = 6, 5, 4
    chol = np.tril(rng.normal(size=(B, D, D)) * 0.3)
    chol[:, range(D), range(D)] = np.abs(chol[:, range(D), range(D)]) + 0.7
    mean = rng.normal(size=(B, D))
    x = rng.normal(size=(n, B, D))
    mt = _T(torch, dev, mean).requires_grad_(True)
    ct = _T(torch, dev, chol).requires_grad_(True)
    xt = _T(torch, dev, x).requires_grad_(True)
    w = _T(torch, dev, rng.normal(size=(n, B)))
    lp = zs.distributions.MultivariateNormalCholesky(mt, ct).log_prob(xt)
    (lp * w).sum().backward()
    m2 = mt.detach().double().requires_grad_(True)
    c2 = ct.detach().double().requires_grad_(True)
    x2 = xt.detach().double().requires_grad_(True)
    ref = torch.distributions.MultivariateNormal(m2, scale_tril=c2).log_prob(x2)
    np.testing.assert_allclose(lp.detach().cpu().numpy(),
                               ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    (ref * w.double()).sum().backward()
    for got, want in ((xt.grad, x2.grad), (mt.grad, m2.grad),
                      (ct.grad, torch.tril(c2.grad))):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(),
                                   rtol=2e-3, atol=2e-3)
    # given broadcast INTO the batch axes
    x1 = _T(torch, dev, x[0, 0])
    lp1 = zs.distributions.MultivariateNormalCholesky(
        mt.detach(), ct.detach()).log_prob(x1)
    ref1 = torch.distributions.MultivariateNormal(
        m2.detach(), scale_tril=c2.detach()).log_prob(x1.double())
    np.testing.assert_allclose(lp1.cpu().numpy(), ref1.cpu().numpy(),
                               rtol=1e-4, atol=1e-4)


def test_sample_matches_oracle_stream_and_moments(env):
    zs, torch, dev = env
    g = np.load(GOLD)
    mean, chol, cov = g['s23_mean'], g['s23_chol'], g['s23_cov']
    zs.set_random_seed(77)
    d = zs.distributions.MultivariateNormalCholesky(
        _T(torch, dev, mean), _T(torch, dev, chol))
    from zhusuan_amd import utils
    seed, off = utils._state.seed, utils._state.op_counter
    s = d.sample(37).cpu().numpy()
    assert s.shape == (37, 10, 11, 3)
    want = R.MultivariateNormalCholesky(mean, chol).sample(37, seed=seed,
                                                           offset=off)
    np.testing.assert_allclose(s, want, rtol=1e-5, atol=2e-5)
    assert tuple(d.sample().shape) == (10, 11, 3)
    # test_multivariate.py:73-94
    big = d.sample(20000).cpu().numpy()
    np.testing.assert_allclose(big.mean(0), mean, rtol=5e-2, atol=5e-2)
    for i in range(0, 10, 3):
        for j in range(0, 11, 5):
            np.testing.assert_allclose(np.cov(big[:, i, j, :].T), cov[i, j],
                                       rtol=1e-1, atol=1e-1)


def test_sample_reparameterized(env):
    """test_multivariate.py:122-136."""
    zs, torch, dev = env
    g = np.load(GOLD)
    mean = _T(torch, dev, g['s23_mean']).requires_grad_(True)
    chol = _T(torch, dev, g['s23_chol']).requires_grad_(True)
    s = zs.distributions.MultivariateNormalCholesky(mean, chol).sample(3)
    gm, gc = torch.autograd.grad(s.sum(), [mean, chol])
    assert gm is not None and gc is not None
    np.testing.assert_allclose(gm.cpu().numpy(), 3.0)
    s = zs.distributions.MultivariateNormalCholesky(
        mean, chol, is_reparameterized=False).sample(3)
    assert not s.requires_grad


def test_hmc_with_mvn_prior_recovers_covariance(env):
    """A correlated Gaussian target written with bn.multivariate_normal_cholesky
    (bn.py:840-870) sampled by the generic HMC plan."""
    zs, torch, dev = env
    rng = np.random.RandomState(0)
    D, C = 6, 4000
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + 0.3 * np.eye(D)
    chol = np.linalg.cholesky(cov)
    mean = rng.normal(size=D)
    mt, ct = _T(torch, dev, mean), _T(torch, dev, chol)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.multivariate_normal_cholesky('x', mt, ct, n_samples=C)
        return bn

    zs.set_random_seed(5)
    hmc = zs.HMC(step_size=0.1, n_leapfrogs=8, adapt_step_size=True,
                 target_acceptance_rate=0.8)
    x = torch.zeros(C, D, device=dev)
    op, info = hmc.sample(model(), {}, {'x': x})
    for i in range(60):
        op.run()
    assert 0.6 < float(info.acceptance_rate.mean()) < 0.95
    s = x.cpu().numpy()
    np.testing.assert_allclose(s.mean(0), mean, atol=0.08)
    np.testing.assert_allclose(np.cov(s.T), cov, atol=0.12)


def test_zero_rows(env):
    zs, torch, dev = env
    d = zs.distributions.MultivariateNormalCholesky(
        torch.zeros(3, device=dev), torch.eye(3, device=dev))
    lp = d.log_prob(torch.zeros(0, 3, device=dev))
    assert tuple(lp.shape) == (0,)
    assert tuple(d.sample(0).shape) == (0, 3)
