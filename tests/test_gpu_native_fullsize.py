"""Plan-level parity of the NATIVE dense-likelihood plans at the sizes
BASELINE.json names, and free-running.

  * configs[2] (Bayesian logistic regression, 10^6 x 256, 32 768 chains) and
    configs[4] (topic-model E step, chain axes [8 192, 5 000], K = 128,
    V = 12 419): ONE committed transition of the product's HMC on the native
    plan -- model written with the reference's literal dense spelling -- at
    the full size, compared chain by chain with the oracle
    (oracle/hmc_ref.py: momentum, leapfrog schedule, Hamiltonians, MH test)
    on a spread subset of chains.  Chains are independent and the random
    stream is keyed by the GLOBAL chain index, so the oracle reproduces any
    subset exactly; its likelihood is evaluated in float64 for those rows
    only.  At configs[2] the energies are ~6e5, where float32 (the
    reference's own precision, hmc.py:22) resolves 1/16: the Hamiltonians are
    held to a few float32 ulps, the accepted states to 2e-5, the MH decision
    to the device's OWN published acceptance exactly and to the oracle's
    wherever the two acceptances do not straddle the uniform.
  * free-running device vs oracle on identical seeds for the two families
    (north_star: acceptance and ESS within 1 %): 50 adaptive + 300 recorded
    transitions, nothing re-synchronised, step size, mean acceptance and mean
    reference-estimator ESS within 1 %.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32 = np.float32


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


# -- float64-likelihood oracle models on an explicit set of chain rows ---------
# every test on the exact-fp32 MFMA kernels and on the bf16x3 ones
# (csrc/b3_kernel.h), at the SAME tolerances
ARITH = pytest.mark.parametrize('arith', ['fp32', 'bf16x3'])


def _used(hmc, arith):
    assert hmc.likelihood_arithmetic_used == arith, (
        hmc.likelihood_arithmetic_used, arith)


def _normal_prior(q, mean, logstd):
    """(log N(q), d/dq) in float64, summed over the last axis
    (univariate.py:174-181, group_ndims = 1)."""
    prec = np.exp(-2.0 * logstd)
    lp = np.sum(-0.5 * np.log(2 * np.pi) - logstd -
                0.5 * prec * (q - mean) ** 2, axis=-1)
    return lp, -prec * (q - mean)


def blr_rows_model(X64, y64):
    """w ~ N(0, 1), y ~ Bernoulli(w X^T) (univariate.py:398-403)."""
    def parts(w):
        w = w.astype(np.float64)
        l = w @ X64.T
        ll = (y64 * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum(-1)
        g = (y64 - 1.0 / (1.0 + np.exp(-l))) @ X64
        lp, gp = _normal_prior(w, 0.0, np.zeros(w.shape[-1]))
        return ll + lp, g + gp
    return (lambda qs: parts(qs[0])[0].astype(F32),
            lambda qs: [parts(qs[0])[1].astype(F32)])


def lntm_rows_model(phi64, x_rows64, mean_rows64, logstd64):
    """eta ~ N(mean[doc], exp(logstd)), x ~ UnnormalizedMultinomial(
    log(softmax(eta) . phi)) per (chain, document) row (lntm_mcem.py:33-48,
    multivariate.py:435-443)."""
    def parts(eta):
        eta = eta.astype(np.float64)
        t = np.exp(eta - eta.max(-1, keepdims=True))
        theta = t / t.sum(-1, keepdims=True)
        s = theta @ phi64
        ll = (x_rows64 * np.log(s)).sum(-1)
        g_theta = (x_rows64 / s) @ phi64.T
        g_eta = theta * (g_theta - (g_theta * theta).sum(-1, keepdims=True))
        lp, gp = _normal_prior(eta, mean_rows64, logstd64)
        return ll + lp, g_eta + gp
    return (lambda qs: parts(qs[0])[0].astype(F32),
            lambda qs: [parts(qs[0])[1].astype(F32)])


def _compare_subset(info_sub, x_dev, x_before, rinfo, x_ref, ref, ulp,
                    q_rtol=2e-5):
    """Device vs oracle for the chains of a subset.  `ulp`: float32 spacing
    at the magnitude of the energies."""
    acc_d = info_sub['acceptance_rate'].astype(np.float64)
    h0_d, h1_d = (info_sub[k].astype(np.float64)
                  for k in ('orig_hamiltonian', 'hamiltonian'))
    acc_r = np.asarray(rinfo.acceptance_rate, np.float64).reshape(-1)
    h0_r = np.asarray(rinfo.orig_hamiltonian, np.float64).reshape(-1)
    h1_r = np.asarray(rinfo.hamiltonian, np.float64).reshape(-1)
    # log-densities and Hamiltonians to a few float32 ulps of their magnitude
    np.testing.assert_allclose(info_sub['orig_log_prob'],
                               np.asarray(rinfo.orig_log_prob).reshape(-1),
                               rtol=0, atol=4 * ulp)
    np.testing.assert_allclose(h0_d, h0_r, rtol=0, atol=4 * ulp)
    np.testing.assert_allclose(h1_d, h1_r, rtol=0, atol=6 * ulp)
    # the device's acceptance is exp(min(dH, 0)) of ITS published energies
    # (hmc.py:54-55); float32 subtraction of the two, so one more ulp
    np.testing.assert_allclose(
        acc_d, np.exp(np.minimum(h0_d - h1_d, 0.0)), rtol=0,
        atol=2e-6 + 1.5 * ulp)
    # ... and within what the energy tolerance allows of the oracle's
    d_dh = np.abs((h0_d - h1_d) - (h0_r - h1_r))
    assert d_dh.max() <= 8 * ulp, d_dh.max()
    # MH decisions: exactly `u < acc` (strict, hmc.py:486) on the device's own
    # acceptance ...
    u = np.asarray(ref.last_u01, F32).reshape(-1)
    n = acc_d.shape[0]
    moved = (x_dev.reshape(n, -1) != x_before.reshape(n, -1)).any(1)
    np.testing.assert_array_equal(moved, u < info_sub['acceptance_rate'])
    # ... equal to the oracle's unless the two acceptances straddle u
    accept_r = u < np.asarray(rinfo.acceptance_rate, F32).reshape(-1)
    differ = moved != accept_r
    lo, hi = np.minimum(acc_d, acc_r), np.maximum(acc_d, acc_r)
    assert np.all(~differ | ((u >= lo - 1e-7) & (u <= hi + 1e-7)))
    same = ~differ
    xd = x_dev.reshape(n, -1)[same]
    xr = np.asarray(x_ref).reshape(n, -1)[same]
    scale = max(1.0, float(np.abs(xr).max()))
    np.testing.assert_allclose(xd, xr, rtol=0, atol=q_rtol * scale)
    # selected log-prob (hmc.py:490-493)
    lp_r = np.asarray(rinfo.log_prob).reshape(-1)[same]
    np.testing.assert_allclose(info_sub['log_prob'][same], lp_r, rtol=0,
                               atol=6 * ulp)
    return int(differ.sum()), float(d_dh.max())


def _subset_info(info, ids_t):
    return {k: getattr(info, k).reshape(-1)[ids_t].cpu().numpy()
            for k in ('acceptance_rate', 'orig_hamiltonian', 'hamiltonian',
                      'orig_log_prob', 'log_prob')}


@ARITH
def test_config3_full_size_transition_matches_oracle_on_a_subset(env, arith):
    zs, torch, dev = env
    from oracle.hmc_ref import HMC as RefHMC
    C, N, D, L, eps, seed = 32768, 1000000, 256, 10, 2e-4, 31
    # the data BASELINE.md / SURVEY 8d c3 name, as bench.py::extra_config3
    # draws them: X ~ N(0, 1) float32, w* ~ N(0, 1), y ~ Bernoulli(
    # sigmoid(X w* / sqrt(D))) from numpy.random.default_rng(0)
    rng0 = np.random.default_rng(0)
    X_h = rng0.standard_normal((N, D), dtype=np.float32)
    w_h = rng0.standard_normal(D).astype(np.float32)
    y_h = rng0.random(N) < 1.0 / (1.0 + np.exp(-(X_h @ w_h) / F32(D ** 0.5)))
    X = torch.from_numpy(X_h).to(dev)
    w_true = torch.from_numpy(w_h).to(dev)
    y = torch.from_numpy(y_h.astype(np.float32)).to(dev)
    del X_h, y_h
    g = torch.Generator(device=dev).manual_seed(0)
    # chains scattered over the posterior's bulk (width ~2e-3 at N = 10^6)
    w = (w_true / D ** 0.5 +
         2e-3 * torch.randn(C, D, device=dev, generator=g)).contiguous()

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        wn = bn.normal('w', torch.zeros(D, device=dev), std=1., n_samples=C,
                       group_ndims=1)
        # the reference's literal spelling: [32 768, 10^6] logits = 131 GB if
        # it were ever materialised
        bn.bernoulli('y', wn.tensor @ X.t(), group_ndims=1,
                     dtype=torch.float32)
        return bn

    rng = np.random.RandomState(0)
    ids = np.unique(np.concatenate([
        [0, 1, 31, 32, 63, 64, C // 2, C - 65, C - 64, C - 1],
        rng.randint(0, C, size=6)])).astype(np.int64)
    ids_t = torch.tensor(ids, device=dev)
    w_before = w[ids_t].cpu().numpy()
    hmc = zs.HMC(step_size=eps, n_leapfrogs=L, seed=seed,
                 likelihood_arithmetic=arith)
    op, info = hmc.sample(blr(), {'y': y}, {'w': w})
    assert hmc.plan_kind == 'linear_bernoulli'
    _used(hmc, arith)
    op.run()

    model = blr_rows_model(X.cpu().numpy().astype(np.float64),
                           y.cpu().numpy().astype(np.float64))
    xr = w_before.copy()
    ref = RefHMC(step_size=eps, n_leapfrogs=L, seed=seed)
    ref.sample(model[0], model[1], [xr], chain_offset=ids.astype(np.uint64))
    rinfo = ref.step()
    h_scale = float(np.abs(rinfo.orig_hamiltonian).max())
    ulp = float(np.spacing(F32(h_scale)))
    differ, d_dh = _compare_subset(_subset_info(info, ids_t),
                                   w[ids_t].cpu().numpy(), w_before, rinfo,
                                   xr, ref, ulp)
    assert differ <= 2
    # every chain ran: the transition moved the population as the subset
    a_all = float(info.acceptance_rate.mean().item())
    assert 0.3 < a_all <= 1.0
    assert abs(a_all - float(np.mean(rinfo.acceptance_rate))) < 0.25


@ARITH
def test_config5_full_size_transition_matches_oracle_on_a_subset(env, arith):
    zs, torch, dev = env
    from oracle.hmc_ref import HMC as RefHMC
    n_docs, K, V, L, eps, seed = 5000, 128, 12419, 3, 0.04, 32
    free_b, _ = torch.cuda.mem_get_info()
    n_chains = 8192
    # latent + p, q_new, grad, operand (+ the evaluation of the first run)
    while n_chains > 64 and 7.0 * n_chains * n_docs * K * 4 > free_b:
        n_chains //= 2
    g = torch.Generator(device=dev).manual_seed(0)
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    mix = torch.softmax(torch.randn(n_docs, K, device=dev, generator=g), -1)
    words = torch.multinomial(mix @ phi, 1000, replacement=True, generator=g)
    x = torch.zeros(n_docs, V, device=dev).scatter_add_(
        1, words, torch.ones(words.shape, device=dev))
    del mix, words
    eta_mean = 0.1 * torch.randn(n_docs, K, device=dev, generator=g)
    eta_logstd = 0.1 * torch.randn(K, device=dev, generator=g)
    eta = torch.empty(n_chains, n_docs, K, device=dev)
    eta.normal_(0.0, 0.3, generator=g)

    @zs.meta_bayesian_net()
    def lntm():
        bn = zs.BayesianNet()
        e = bn.normal('eta', eta_mean, logstd=eta_logstd, n_samples=n_chains,
                      group_ndims=1)
        theta = torch.softmax(e.tensor, -1)                 # lntm_mcem.py:39
        pred = (theta.reshape(-1, K) @ phi).reshape(n_chains, n_docs, V)
        bn.unnormalized_multinomial('x', torch.log(pred),
                                    normalize_logits=False,
                                    dtype=torch.float32)
        return bn

    rows = n_chains * n_docs
    rng = np.random.RandomState(1)
    ids = np.unique(np.concatenate([
        [0, 1, 63, 64, n_docs - 1, n_docs, rows // 2, rows - n_docs,
         rows - 65, rows - 1], rng.randint(0, rows, size=38)])).astype(np.int64)
    ids_t = torch.tensor(ids, device=dev)
    flat = eta.view(rows, K)
    eta_before = flat[ids_t].cpu().numpy()
    hmc = zs.HMC(step_size=eps, n_leapfrogs=L, seed=seed,
                 likelihood_arithmetic=arith)
    op, info = hmc.sample(lntm(), {'x': x}, {'eta': eta})
    assert hmc.plan_kind == 'mixture_multinomial'
    if n_chains >= 128:
        _used(hmc, arith)
    op.run()

    docs = ids % n_docs
    model = lntm_rows_model(
        phi.cpu().numpy().astype(np.float64),
        x[torch.tensor(docs, device=dev)].cpu().numpy().astype(np.float64),
        eta_mean[torch.tensor(docs, device=dev)].cpu().numpy().astype(
            np.float64),
        eta_logstd.cpu().numpy().astype(np.float64))
    xr = eta_before.copy()
    ref = RefHMC(step_size=eps, n_leapfrogs=L, seed=seed)
    ref.sample(model[0], model[1], [xr], chain_offset=ids.astype(np.uint64))
    rinfo = ref.step()
    h_scale = float(np.abs(rinfo.orig_hamiltonian).max())
    ulp = float(np.spacing(F32(h_scale)))
    differ, d_dh = _compare_subset(_subset_info(info, ids_t),
                                   flat[ids_t].cpu().numpy(), eta_before,
                                   rinfo, xr, ref, ulp)
    assert differ <= 2
    a_all = float(info.acceptance_rate.mean().item())
    assert abs(a_all - float(np.mean(rinfo.acceptance_rate))) < 0.15
    assert n_chains == 8192 or free_b < 7.0 * 8192 * n_docs * K * 4


# -- free-running, device vs oracle on identical seeds ------------------------
def _free_run(zs, torch, hmc, op, info, q_dev, flags, ref, q_ref, n_adapt,
              n_draws):
    from oracle import ess_ref
    rows = int(np.prod(q_ref.shape[:-1]))
    D = q_ref.shape[-1]
    rec_g = torch.empty((n_draws, rows, D), device=q_dev.device)
    rec_r = np.empty((n_draws, rows, D), np.float32)
    acc_g = torch.zeros((), device=q_dev.device)
    acc_r = 0.0
    for i in range(n_adapt + n_draws):
        a = i < n_adapt
        rinfo = ref.step(adapt_step_size=a,
                         adapt_mass=a if len(flags) > 1 else None)
        op.run(feed_dict={f: a for f in flags}, sync=False)
        if not a:
            j = i - n_adapt
            rec_g[j].copy_(q_dev.view(rows, D))
            rec_r[j] = q_ref.reshape(rows, D)
            acc_g += info.acceptance_rate.mean()
            acc_r += float(np.mean(rinfo.acceptance_rate))
    hmc.check_numerics()
    ess_g = zs.diagnostics.effective_sample_size_device(rec_g, burn_in=0)
    ess_g = float(ess_g[torch.isfinite(ess_g)].mean().item())
    ess_r = np.array([ess_ref.effective_sample_size(rec_r[:, c, :], burn_in=0)
                      for c in range(rows)])
    ess_r = float(ess_r[np.isfinite(ess_r)].mean())
    return ((float(acc_g.item()) / n_draws, ess_g,
             float(info.updated_step_size.item())),
            (acc_r / n_draws, ess_r, float(ref.step_size)))


def _within_one_percent(got, want):
    for g, w, what in zip(got, want, ('acceptance', 'ESS', 'step size')):
        assert abs(g - w) <= 0.01 * abs(w), (what, got, want)


@ARITH
def test_free_running_logistic_regression_within_one_percent(env, arith):
    """The configs[2] family on the native plan, literal spelling, 1 536
    chains x 24 weights x 600 rows, step-size adaptation for the first 50 of
    350 transitions."""
    zs, torch, dev = env
    from helpers_hmc_cases import blr_model
    from oracle.hmc_ref import HMC as RefHMC
    rng = np.random.RandomState(5)
    N, D, C = 600, 24, 1536
    X = rng.normal(size=(N, D)).astype(F32)
    wt = rng.normal(size=D).astype(F32)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ wt))).astype(np.int32)
    w0 = (0.1 * rng.normal(size=(C, D))).astype(F32)
    kw = dict(step_size=0.02, n_leapfrogs=6, target_acceptance_rate=0.8,
              seed=41)
    lj, grad = blr_model(X, y)
    xr = w0.copy()
    ref = RefHMC(adapt_step_size=True, **kw)
    ref.sample(lj, grad, [xr])
    Xt, yt = torch.tensor(X, device=dev), torch.tensor(y, device=dev)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D, device=dev), std=1., n_samples=C,
                      group_ndims=1)
        bn.bernoulli('y', w.tensor @ Xt.t(), group_ndims=1)
        return bn
    flag = zs.placeholder(bool)
    hmc = zs.HMC(adapt_step_size=flag, likelihood_arithmetic=arith, **kw)
    w = torch.tensor(w0, device=dev)
    op, info = hmc.sample(blr(), {'y': yt}, {'w': w})
    assert hmc.plan_kind == 'linear_bernoulli'
    _used(hmc, arith)
    got, want = _free_run(zs, torch, hmc, op, info, w, (flag,), ref, xr, 50,
                          300)
    _within_one_percent(got, want)
    assert 0.6 < got[0] < 0.95


@ARITH
def test_free_running_topic_model_within_one_percent(env, arith):
    """The configs[4] family on the native plan, literal spelling, E-step
    objective, chain axes [256, 16] (4 096 rows: the ESS estimator's mean over
    rows is then good to well under 1 % although free-running chains part
    ways at the first borderline accept decision), K = 12, V = 80, step-size
    AND mass adaptation for the first 50 of 350 transitions."""
    zs, torch, dev = env
    from helpers_hmc_cases import lntm_model
    from oracle.hmc_ref import HMC as RefHMC
    rng = np.random.RandomState(6)
    n_chains, n_docs, K, V = 256, 16, 12, 80
    beta = rng.normal(size=(K, V)).astype(F32)
    x = rng.poisson(1.5, size=(n_docs, V)).astype(F32)
    eta_mean = (0.3 * rng.normal(size=K)).astype(F32)
    eta_logstd = (0.2 * rng.normal(size=K)).astype(F32)
    eta0 = (0.5 * rng.normal(size=(n_chains, n_docs, K))).astype(F32)
    kw = dict(step_size=5e-3, n_leapfrogs=5, target_acceptance_rate=0.6,
              mass_collect_iters=10, seed=42)
    lj, grad = lntm_model(beta, x, eta_mean, eta_logstd)
    xr = eta0.copy()
    ref = RefHMC(adapt_step_size=True, adapt_mass=True, **kw)
    ref.sample(lj, grad, [xr])
    T = lambda a: torch.tensor(a, device=dev)
    beta_t, x_t, mean_t, logstd_t = T(beta), T(x), T(eta_mean), T(eta_logstd)

    @zs.meta_bayesian_net(scope='lntm')
    def lntm():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', mean_t.unsqueeze(0).repeat(n_docs, 1),
                        logstd=logstd_t, n_samples=n_chains, group_ndims=1)
        theta = torch.softmax(eta.tensor, -1)
        b = bn.normal('beta', torch.zeros(K, V, device=dev), logstd=10.0,
                      group_ndims=1)
        phi = torch.softmax(b.tensor, -1)
        pred = (theta.reshape(-1, K) @ phi).reshape(n_chains, n_docs, V)
        bn.unnormalized_multinomial('x', torch.log(pred),
                                    normalize_logits=False,
                                    dtype=torch.float32)
        return bn
    model = lntm()
    model.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
    f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
    hmc = zs.HMC(adapt_step_size=f_ss, adapt_mass=f_m,
                 likelihood_arithmetic=arith, **kw)
    eta = T(eta0)
    op, info = hmc.sample(model, {'x': x_t, 'beta': beta_t}, {'eta': eta})
    assert hmc.plan_kind == 'mixture_multinomial'
    _used(hmc, arith)
    got, want = _free_run(zs, torch, hmc, op, info, eta, (f_ss, f_m), ref, xr,
                          50, 300)
    _within_one_percent(got, want)
    assert 0.4 < got[0] < 0.9
