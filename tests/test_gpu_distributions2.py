"""GPU parity of Laplace / Gamma / InverseGamma / Beta
(csrc/distributions2.hip through zhusuan_amd.distributions) against the
reference's own test vectors (scipy targets, as in the reference's
tests/distributions/test_univariate.py), the oracle's analytic gradients, the
sampling moments, and HMC with a Gamma / Beta model through the BayesianNet
front-end."""
import numpy as np
import pytest
from scipy import stats

from oracle import distributions_ref as dref
from test_oracle_distributions2 import VECTORS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


@pytest.mark.parametrize('name', sorted(VECTORS))
def test_log_prob_reference_vectors(env, name):
    zs, torch, dev = env
    target, vecs = VECTORS[name]
    for a, b, given in vecs:
        a32, b32, g32 = (np.array(v, np.float32) for v in (a, b, given))
        d = getattr(zs.distributions, name)(torch.tensor(a32, device=dev),
                                            torch.tensor(b32, device=dev))
        got = d.log_prob(torch.tensor(g32, device=dev)).cpu().numpy()
        want = target(g32, a32, b32)
        assert got.shape == want.shape
        # float32 lgamma / log on the device vs float64 scipy
        np.testing.assert_allclose(got, want, rtol=3e-5,
                                   atol=3e-6 * max(1.0, np.abs(want).max()))
        np.testing.assert_allclose(
            d.prob(torch.tensor(g32, device=dev)).cpu().numpy(), np.exp(want),
            rtol=1e-3, atol=1e-30)


@pytest.mark.parametrize('name', sorted(VECTORS))
@pytest.mark.parametrize('group_ndims', [0, 1, 2])
def test_gradients_and_group_ndims(env, name, group_ndims):
    zs, torch, dev = env
    rng = np.random.RandomState(7)
    a = rng.uniform(0.6, 4.0, size=(6,)).astype(np.float32)       # ROW param
    b = rng.uniform(0.6, 3.0, size=(3, 1, 6)).astype(np.float32)  # needs expand
    x = (rng.uniform(0.1, 0.9, size=(3, 5, 6)) if name == 'Beta' else
         rng.uniform(0.3, 4.0, size=(3, 5, 6))).astype(np.float32)
    at = torch.tensor(a, device=dev, requires_grad=True)
    bt = torch.tensor(b, device=dev, requires_grad=True)
    xt = torch.tensor(x, device=dev, requires_grad=True)
    d = getattr(zs.distributions, name)(at, bt, group_ndims=group_ndims)
    lp = d.log_prob(xt)
    ref = getattr(dref, name)(a, b, group_ndims=group_ndims)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), ref.log_prob(x),
                               rtol=3e-5, atol=3e-5)
    coef = torch.linspace(0.5, 1.5, lp.numel(), device=dev).reshape(lp.shape)
    (lp * coef).sum().backward()
    gx, ga, gb = ref.grads(x)
    c = coef.cpu().numpy().astype(np.float64)
    c = c.reshape(c.shape + (1,) * (x.ndim - c.ndim))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gx * c, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(at.grad.cpu().numpy(),
                               (ga * c).sum((0, 1)), rtol=3e-4, atol=3e-3)
    np.testing.assert_allclose(bt.grad.cpu().numpy(),
                               (gb * c).sum(1, keepdims=True), rtol=3e-4,
                               atol=3e-3)


def test_constructor_errors(env):
    zs, torch, dev = env
    with pytest.raises(ValueError, match='should be broadcastable'):
        zs.distributions.Gamma(torch.ones(2, device=dev), torch.ones(3, device=dev))
    with pytest.raises(TypeError, match='same dtype'):
        zs.distributions.Beta(torch.ones(2, device=dev),
                              torch.ones(2, device=dev, dtype=torch.float64))
    g = zs.distributions.Gamma(torch.ones(2, 1, device=dev),
                               torch.ones(3, device=dev))
    assert tuple(g.batch_shape) == (2, 3) and not g.is_reparameterized
    assert zs.distributions.Laplace(0., 1.).is_reparameterized


SAMPLERS = {
    'Laplace': ((0.7, 1.3), stats.laplace(0.7, scale=1.3)),
    'Gamma': ((2.5, 1.7), stats.gamma(2.5, scale=1 / 1.7)),
    'GammaSmall': ((0.4, 2.0), stats.gamma(0.4, scale=0.5)),
    'InverseGamma': ((4.5, 2.0), stats.invgamma(4.5, scale=2.0)),
    'Beta': ((2.0, 5.0), stats.beta(2.0, 5.0)),
    'BetaU': ((0.5, 0.5), stats.beta(0.5, 0.5)),
}


@pytest.mark.parametrize('case', sorted(SAMPLERS))
def test_sampling_distribution(env, case):
    """Kolmogorov-Smirnov against the exact CDF (2e5 draws)."""
    zs, torch, dev = env
    (a, b), dist = SAMPLERS[case]
    name = case.replace('Small', '').replace('U', '')
    zs.set_random_seed(1234)
    d = getattr(zs.distributions, name)(torch.full((4,), a, device=dev),
                                        torch.tensor(b, device=dev))
    s = d.sample(50000)
    assert tuple(s.shape) == (50000, 4)
    x = s.cpu().numpy().reshape(-1).astype(np.float64)
    assert np.isfinite(x).all()
    ks = stats.kstest(x, dist.cdf)
    assert ks.statistic < 0.006, ks
    # a second call continues the stream (different numbers)
    assert not torch.equal(s, d.sample(50000))


def test_hmc_gamma_beta_model(env):
    """lam ~ Gamma(3, 2), th ~ Beta(2, 4) sampled on the real line (u0 = log
    lam, u1 = logit th, with the Jacobians): the generic HMC plan
    differentiates through the HIP log_prob ops; moments vs the analytic
    ones."""
    zs, torch, dev = env
    C = 2048

    def log_joint(obs):
        u = obs['u']
        lam = torch.exp(u[:, 0])
        th = torch.sigmoid(u[:, 1])
        g = zs.distributions.Gamma(torch.tensor(3.0, device=dev),
                                   torch.tensor(2.0, device=dev))
        b = zs.distributions.Beta(torch.tensor(2.0, device=dev),
                                  torch.tensor(4.0, device=dev))
        return (g.log_prob(lam) + u[:, 0] +
                b.log_prob(th) + torch.log(th) + torch.log1p(-th))

    u = torch.zeros(C, 2, device=dev)
    hmc = zs.HMC(step_size=0.1, n_leapfrogs=8, adapt_step_size=True,
                 target_acceptance_rate=0.8, seed=3)
    op, info = hmc.sample(log_joint, {}, {'u': u})
    for _ in range(150):
        op.run()
    lam, th = [], []
    for _ in range(60):
        op.run()
        lam.append(torch.exp(u[:, 0]).cpu().numpy())
        th.append(torch.sigmoid(u[:, 1]).cpu().numpy())
    lam, th = np.concatenate(lam), np.concatenate(th)
    assert abs(lam.mean() - 1.5) < 0.03 and abs(lam.var() - 0.75) < 0.05
    assert abs(th.mean() - 1 / 3) < 0.01 and abs(th.var() - 2 / 63) < 0.004
