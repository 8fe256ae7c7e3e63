"""The HMC cases of tests/golden/hmc_reference_traces.npz (produced by
running the reference's own zhusuan/hmc.py, see oracle/make_golden_hmc.py),
restated for the oracle (NumPy log-joint + analytic gradient)."""
import numpy as np

from oracle import distributions_ref as dref
from oracle import pmf_ref
from oracle.hmc_case_data import (blr_bias_data, blr_data, lntm_data,
                                  lntm_ragged_data, pmf_data,
                                  softmax_regression_data)

F32 = np.float32


def gaussian_model(mean, logstd):
    mean, logstd = mean.astype(F32), logstd.astype(F32)
    c = F32(-0.5 * np.log(2 * np.pi))
    prec = np.exp(F32(-2) * logstd).astype(F32)

    def log_joint(qs):
        x = qs[0]
        return np.sum(c - logstd - F32(0.5) * prec * np.square(x - mean),
                      axis=-1, dtype=F32)

    def grad(qs):
        return [(-prec * (qs[0] - mean)).astype(F32)]
    return log_joint, grad


def coupled_model(prec_x):
    px = prec_x.astype(F32)

    def log_joint(qs):
        x, y = qs
        sx, sy = x.sum(-1, dtype=F32), y.sum(-1, dtype=F32)
        return (F32(-0.5) * np.sum(px * np.square(x), axis=-1, dtype=F32)
                - F32(0.5) * np.sum(np.square(y), axis=-1, dtype=F32)
                - F32(0.01) * np.square(sx) * np.square(sy)).astype(F32)

    def grad(qs):
        x, y = qs
        sx, sy = x.sum(-1, dtype=F32), y.sum(-1, dtype=F32)
        gx = -px * x - (F32(0.02) * sx * np.square(sy))[..., None]
        gy = -y - (F32(0.02) * np.square(sx) * sy)[..., None]
        return [gx.astype(F32), gy.astype(F32)]
    return log_joint, grad


def blr_model(X, y):
    """w ~ Normal(0, std = 1) (group_ndims 1), y ~ Bernoulli(w X^T)
    (group_ndims 1): log-joint and its gradient through the matmul."""
    X = X.astype(F32)

    def parts(w):
        prior = dref.Normal(F32(0), std=F32(1), group_ndims=1)
        lik = dref.Bernoulli((w @ X.T).astype(F32), group_ndims=1)
        return prior, lik

    def log_joint(qs):
        prior, lik = parts(qs[0])
        return (prior.log_prob(qs[0]) + lik.log_prob(y)).astype(F32)

    def grad(qs):
        prior, lik = parts(qs[0])
        return [(prior.grad_given(qs[0]) +
                 lik.grad_logits(y) @ X).astype(F32)]
    return log_joint, grad


def lntm_model(beta, x, eta_mean, eta_logstd):
    """The E-step objective of lntm_mcem.py:97-102: log N(eta) +
    sum_v x_v log(softmax(eta) . softmax(beta))_v, chain axes [chains, docs]."""
    e = np.exp(beta - beta.max(-1, keepdims=True)).astype(F32)
    phi = (e / e.sum(-1, keepdims=True)).astype(F32)

    def softmax(eta):
        t = np.exp(eta - eta.max(-1, keepdims=True)).astype(F32)
        return (t / t.sum(-1, keepdims=True)).astype(F32)

    def parts(eta):
        theta = softmax(eta)
        prior = dref.Normal(eta_mean, logstd=eta_logstd, group_ndims=1)
        lik = dref.UnnormalizedMultinomial(
            np.log(theta @ phi).astype(F32), normalize_logits=False,
            dtype=np.float32)
        return theta, prior, lik

    def log_joint(qs):
        _, prior, lik = parts(qs[0])
        return (prior.log_prob(qs[0]) + lik.log_prob(x)).astype(F32)

    def grad(qs):
        theta, prior, lik = parts(qs[0])
        # d/d logits = x; through log, the mixture and the softmax
        g_theta = ((lik.grad_logits(x) / (theta @ phi)) @ phi.T).astype(F32)
        g_eta = theta * (g_theta - (g_theta * theta).sum(-1, keepdims=True))
        return [(prior.grad_given(qs[0]) + g_eta).astype(F32)]
    return log_joint, grad


def softmax_regression_model(X, y):
    """w[k, f] ~ Normal(0, 1) (group_ndims 2), y ~ Categorical(logits[n, k] =
    <X[n], w[k]>) (group_ndims 1)."""
    X = X.astype(F32)

    def parts(w):
        prior = dref.Normal(F32(0), std=F32(1), group_ndims=2)
        logits = np.einsum('nf,ckf->cnk', X, w).astype(F32)
        return prior, dref.Categorical(logits, group_ndims=1)

    def log_joint(qs):
        prior, lik = parts(qs[0])
        return (prior.log_prob(qs[0]) + lik.log_prob(y)).astype(F32)

    def grad(qs):
        prior, lik = parts(qs[0])
        g_logits = lik.grad_logits(y)                       # [C, N, K]
        return [(prior.grad_given(qs[0]) +
                 np.einsum('cnk,nf->ckf', g_logits, X)).astype(F32)]
    return log_joint, grad


def pmf_model(su, sv, r, v_obs, alphas):
    """pmf_hmc.py:19-31,135-141 with the item factors observed: HMC over the
    user factors u[particles, users, factors]."""
    def log_joint(qs):
        return pmf_ref.log_joint(qs[0], v_obs, su, sv, r, *alphas)

    def grad(qs):
        return [pmf_ref.grad_log_joint(qs[0], v_obs, su, sv, r, *alphas)[0]]
    return log_joint, grad


def cases():
    D = 10
    stdev = (1.0 / (np.arange(D) + 1)).astype(F32)
    yield dict(name='gauss_adapt', latent_names=['x'],
               model=gaussian_model(np.zeros(D, F32), np.log(stdev).astype(F32)),
               params=dict(mean=np.zeros(D, F32), logstd=np.log(stdev).astype(F32)),
               hmc_kwargs=dict(step_size=1e-3, n_leapfrogs=5,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.9),
               n_iters=22, flags=lambda i: (i < 16, i < 16), seed=11)
    yield dict(name='coupled', latent_names=['x', 'y'],
               model=coupled_model(np.linspace(0.5, 2.0, 6).astype(F32)),
               params=dict(prec_x=np.linspace(0.5, 2.0, 6).astype(F32)),
               hmc_kwargs=dict(step_size=0.08, n_leapfrogs=7),
               n_iters=6, flags=lambda i: (None, None), seed=12)
    D = 33
    yield dict(name='gauss_ss', latent_names=['x'],
               model=gaussian_model(np.linspace(-1, 1, D).astype(F32),
                                    np.linspace(-0.7, 0.4, D).astype(F32)),
               params=dict(mean=np.linspace(-1, 1, D).astype(F32),
                           logstd=np.linspace(-0.7, 0.4, D).astype(F32)),
               hmc_kwargs=dict(step_size=0.05, n_leapfrogs=4,
                               adapt_step_size=True,
                               target_acceptance_rate=0.8),
               n_iters=15, flags=lambda i: (True, None), seed=13)
    D = 260
    yield dict(name='gauss_ring', latent_names=['x'],
               model=gaussian_model(np.linspace(-2, 2, D).astype(F32),
                                    np.linspace(0.0, 1.2, D).astype(F32)),
               params=dict(mean=np.linspace(-2, 2, D).astype(F32),
                           logstd=np.linspace(0.0, 1.2, D).astype(F32)),
               hmc_kwargs=dict(step_size=0.02, n_leapfrogs=6,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.8,
                               mass_collect_iters=4),
               n_iters=26, flags=lambda i: (i < 22, i < 18), seed=14)
    D = 10
    stdev = (1.0 / (np.arange(D) + 1)).astype(F32)
    yield dict(name='gaussian_py', latent_names=['x'],
               model=gaussian_model(np.zeros(D, F32), np.log(stdev).astype(F32)),
               params=dict(mean=np.zeros(D, F32), std=stdev),
               hmc_kwargs=dict(step_size=1e-3, n_leapfrogs=5,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.9),
               n_iters=30, flags=lambda i: (i < 15, i < 15), seed=1)
    X, y, _ = blr_data()
    yield dict(name='blr', latent_names=['w'], model=blr_model(X, y),
               params=dict(X=X, y=y),
               hmc_kwargs=dict(step_size=0.02, n_leapfrogs=6,
                               adapt_step_size=True,
                               target_acceptance_rate=0.8),
               n_iters=10, flags=lambda i: (True, None), seed=15)
    beta, x, eta_mean, eta_logstd, _ = lntm_data()
    yield dict(name='lntm', latent_names=['eta'],
               model=lntm_model(beta, x, eta_mean, eta_logstd),
               params=dict(beta=beta, x=x, eta_mean=eta_mean,
                           eta_logstd=eta_logstd),
               hmc_kwargs=dict(step_size=5e-3, n_leapfrogs=5,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.6,
                               mass_collect_iters=3),
               n_iters=14, flags=lambda i: (i < 11, i < 9), seed=16)
    Xs, ys, _ = softmax_regression_data()
    yield dict(name='softmax_reg', latent_names=['w'],
               model=softmax_regression_model(Xs, ys),
               params=dict(X=Xs, y=ys),
               hmc_kwargs=dict(step_size=0.03, n_leapfrogs=5,
                               adapt_step_size=True,
                               target_acceptance_rate=0.8),
               n_iters=8, flags=lambda i: (True, None), seed=17)
    su, sv, r, v_obs, _, alphas = pmf_data()
    yield dict(name='pmf', latent_names=['u'],
               model=pmf_model(su, sv, r, v_obs, alphas),
               params=dict(su=su, sv=sv, r=r, v=v_obs),
               hmc_kwargs=dict(step_size=0.4, n_leapfrogs=6),
               n_iters=6, flags=lambda i: (None, None), seed=18)


def blr_bias_model(X1, X2, y):
    """u ~ N(0, 1), v ~ N(0, 0.5^2) (group_ndims 1), b ~ N(0, 2^2) per chain
    (group_ndims 0), y ~ Bernoulli(u X1^T + v X2^T + b[:, None])
    (group_ndims 1): the three-latent model of oracle/make_golden_hmc_r3.py."""
    X1, X2 = X1.astype(F32), X2.astype(F32)

    def parts(u, v, b):
        pri = (dref.Normal(F32(0), std=F32(1), group_ndims=1),
               dref.Normal(F32(0), std=F32(0.5), group_ndims=1),
               dref.Normal(F32(0), std=F32(2)))
        logits = (u @ X1.T + v @ X2.T + b[:, None]).astype(F32)
        return pri, dref.Bernoulli(logits, group_ndims=1)

    def log_joint(qs):
        pri, lik = parts(*qs)
        return (pri[0].log_prob(qs[0]) + pri[1].log_prob(qs[1]) +
                pri[2].log_prob(qs[2]) + lik.log_prob(y)).astype(F32)

    def grad(qs):
        pri, lik = parts(*qs)
        res = lik.grad_logits(y)
        return [(pri[0].grad_given(qs[0]) + res @ X1).astype(F32),
                (pri[1].grad_given(qs[1]) + res @ X2).astype(F32),
                (pri[2].grad_given(qs[2]) + res.sum(-1)).astype(F32)]
    return log_joint, grad


def cases_r3():
    """tests/golden/hmc_reference_traces_r3.npz (oracle/make_golden_hmc_r3.py):
    several latents feeding one dense likelihood."""
    X1, X2, y, _, _, _ = blr_bias_data()
    yield dict(name='blr_bias', latent_names=['u', 'v', 'b'],
               model=blr_bias_model(X1, X2, y),
               params=dict(X1=X1, X2=X2, y=y),
               hmc_kwargs=dict(step_size=0.02, n_leapfrogs=5,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.8,
                               mass_collect_iters=3),
               n_iters=12, flags=lambda i: (i < 10, i < 8), seed=19)
    beta, x, eta_mean, eta_logstd, _ = lntm_ragged_data()
    yield dict(name='lntm_k6', latent_names=['eta'],
               model=lntm_model(beta, x, eta_mean, eta_logstd),
               params=dict(beta=beta, x=x, eta_mean=eta_mean,
                           eta_logstd=eta_logstd),
               hmc_kwargs=dict(step_size=5e-3, n_leapfrogs=5,
                               adapt_step_size=True, adapt_mass=True,
                               target_acceptance_rate=0.6,
                               mass_collect_iters=3),
               n_iters=12, flags=lambda i: (i < 10, i < 8), seed=20)
