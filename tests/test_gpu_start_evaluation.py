"""The carried start evaluation of the native model plans and its public
contract: the reference re-evaluates the joint at
the state a transition starts from on every run (zhusuan/hmc.py:47-50); the
native plans reuse the previous transition's last evaluation instead
(HMC(reuse_start_evaluation=True), the default) -- which is only right while
the model is the same function of the same values.  What the library can see
invalidates it by itself (torch version counters, its own samplers' writes:
zhusuan_amd/_writes.py); what it cannot has `hmc.observed_changed()` /
`hmc.latents_changed()`; and `reuse_start_evaluation=False` is the
reference's behaviour."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _data(seed=0, N=300, D=64, C=96):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(N, D)).astype(np.float32)
    wt = rng.normal(size=D).astype(np.float32)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ wt / 8))).astype(np.int32)
    w0 = (0.05 * rng.normal(size=(C, D))).astype(np.float32)
    return X, y, w0


def _log_joint(w, X, y):
    w = w.astype(np.float64)
    l = w @ X.astype(np.float64).T
    ll = (y * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum(1)
    lp = (-0.5 * np.log(2 * np.pi) - 0.5 * w ** 2).sum(1)
    return ll + lp


def _sampler(zs, torch, dev, X, y, w, **kw):
    D, C = X.shape[1], w.shape[0]

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        wn = bn.normal('w', torch.zeros(D, device=dev), std=1., n_samples=C,
                       group_ndims=1)
        bn.bernoulli('y', wn.tensor @ X.t(), group_ndims=1)
        return bn
    hmc = zs.HMC(step_size=0.02, n_leapfrogs=4, **kw)
    op, info = hmc.sample(blr(), {'y': y}, {'w': w})
    assert hmc.plan_kind == 'linear_bernoulli'
    return hmc, op, info


def _start_matches(info, w_before, X, y):
    want = _log_joint(w_before, X, y)
    got = info.orig_log_prob.cpu().numpy()
    return np.allclose(got, want, rtol=1e-4, atol=5e-3), got, want


def test_observed_written_behind_torch_s_back(env):
    zs, torch, dev = env
    Xh, yh, w0 = _data()
    X = torch.tensor(Xh, device=dev)
    for reuse in (True, False):
        y = torch.tensor(yh, device=dev)
        w = torch.tensor(w0, device=dev)
        hmc, op, info = _sampler(zs, torch, dev, X, y, w, seed=3,
                                 reuse_start_evaluation=reuse)
        for _ in range(3):
            w_before = w.cpu().numpy()
            op.run()
            ok, got, want = _start_matches(info, w_before, Xh, yh)
            assert ok, (reuse, np.abs(got - want).max())
        # the labels flip -- through .data: no version counter moves
        y.data.copy_(1 - y)
        y_new = 1 - yh
        w_before = w.cpu().numpy()
        op.run()
        ok, got, want = _start_matches(info, w_before, Xh, y_new)
        if reuse:
            # (i) nobody told the sampler: it started from the evaluation it
            # carried -- the OLD labels' (this is the documented contract)
            assert not ok
            stale_ok, _, _ = _start_matches(info, w_before, Xh, yh)
            assert stale_ok
            # (ii) told: the next transition evaluates its start
            hmc.observed_changed()
            w_before = w.cpu().numpy()
            op.run()
            ok, got, want = _start_matches(info, w_before, Xh, y_new)
            assert ok, np.abs(got - want).max()
        else:
            # the reference's behaviour: every start evaluated, every run
            # re-reads the observed tensors
            assert ok, np.abs(got - want).max()
        # an in-place torch op is seen either way
        y.copy_(torch.tensor(yh, device=dev))
        w_before = w.cpu().numpy()
        op.run()
        ok, got, want = _start_matches(info, w_before, Xh, yh)
        assert ok, (reuse, np.abs(got - want).max())


def test_reuse_on_and_off_are_bit_identical_when_nothing_changes(env):
    zs, torch, dev = env
    Xh, yh, w0 = _data(seed=1)
    X, y = torch.tensor(Xh, device=dev), torch.tensor(yh, device=dev)
    out = []
    for reuse in (True, False):
        w = torch.tensor(w0, device=dev)
        hmc, op, info = _sampler(zs, torch, dev, X, y, w, seed=5,
                                 reuse_start_evaluation=reuse)
        for _ in range(6):
            op.run()
        out.append((w.cpu().numpy(), info.orig_log_prob.cpu().numpy(),
                    info.acceptance_rate.cpu().numpy()))
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


def test_a_second_sampler_on_the_same_latent_invalidates_the_first(env):
    """every sampler writes latents through the C-ABI,
    which never bumps torch's counters -- two HMC objects (or an HMC and an
    SGMCMC sampler) taking turns on one tensor used to hand each other stale
    start evaluations."""
    zs, torch, dev = env
    Xh, yh, w0 = _data(seed=2)
    X, y = torch.tensor(Xh, device=dev), torch.tensor(yh, device=dev)
    w = torch.tensor(w0, device=dev)
    h1, op1, info1 = _sampler(zs, torch, dev, X, y, w, seed=7)
    h2, op2, info2 = _sampler(zs, torch, dev, X, y, w, seed=8)
    for i in range(4):
        for op, info in ((op1, info1), (op2, info2)):
            w_before = w.cpu().numpy()
            op.run()
            ok, got, want = _start_matches(info, w_before, Xh, yh)
            assert ok, (i, np.abs(got - want).max())
    # ... and an SGLD step in between
    def log_joint(obs):
        wn = obs['w']
        l = wn @ X.t()
        ll = (y * l - torch.clamp(l, min=0) -
              torch.log1p(torch.exp(-l.abs()))).sum(-1)
        return ll - 0.5 * (wn ** 2).sum(-1)
    sgld = zs.SGLD(learning_rate=1e-4, seed=9)
    sop, _ = sgld.sample(log_joint, {}, {'w': w})
    sop.run()
    w_before = w.cpu().numpy()
    op1.run()
    ok, got, want = _start_matches(info1, w_before, Xh, yh)
    assert ok, np.abs(got - want).max()
    # run_many (the C-side loop) marks its writes too
    op2.run_many(3)
    w_before = w.cpu().numpy()
    op1.run()
    ok, got, want = _start_matches(info1, w_before, Xh, yh)
    assert ok, np.abs(got - want).max()


@pytest.mark.parametrize('D', [64, 60])
def test_a_latent_observed_by_another_sampler(env, D):
    """a Gibbs-style alternation -- sampler A's design
    matrix IS sampler B's latent and the other way round.  B moves it through
    the C-ABI (no torch version bump): A must drop the start evaluation it
    carries, and (D = 60: not a kernel width) the zero-padded copy of the
    design matrix it cached, and evaluate its start under the new values."""
    zs, torch, dev = env
    rng = np.random.RandomState(4)
    N, C = 128, 96
    Xt = torch.tensor((0.3 * rng.normal(size=(N, D))).astype(np.float32),
                      device=dev)
    Wt = torch.tensor((0.3 * rng.normal(size=(C, D))).astype(np.float32),
                      device=dev)
    ya = (rng.uniform(size=N) < 0.5).astype(np.int32)
    yb = (rng.uniform(size=C) < 0.5).astype(np.int32)
    ha, opa, infoa = _sampler(zs, torch, dev, Xt, torch.tensor(ya, device=dev),
                              Wt, seed=11)
    hb, opb, infob = _sampler(zs, torch, dev, Wt, torch.tensor(yb, device=dev),
                              Xt, seed=12)
    for i in range(4):
        for op, info, lat, design, y in ((opa, infoa, Wt, Xt, ya),
                                         (opb, infob, Xt, Wt, yb)):
            before, other = lat.cpu().numpy(), design.cpu().numpy()
            op.run()
            ok, got, want = _start_matches(info, before, other, y)
            assert ok, (i, np.abs(got - want).max())
    # the C-side launch loop notes its writes too
    opb.run_many(3)
    before, other = Wt.cpu().numpy(), Xt.cpu().numpy()
    opa.run()
    ok, got, want = _start_matches(infoa, before, other, ya)
    assert ok, np.abs(got - want).max()
