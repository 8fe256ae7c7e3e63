"""Host-side contract of the front-end that needs no GPU: constructor
errors, argument validation, model-structure errors, flags, sharding maths,
ESS (host NumPy in the reference too)."""
import os

import numpy as np
import pytest
import torch

import zhusuan_amd as zs
from zhusuan_amd.distributed import shard_bounds
from zhusuan_amd.hmc import _flag_value


def test_hmc_constructor_contract():
    # hmc.py:270-272
    with pytest.raises(ValueError, match='we should also adapt step size'):
        zs.HMC(adapt_mass=True)
    h = zs.HMC()
    assert (h.n_leapfrogs, h.target_acceptance_rate, h.gamma, h.t0,
            h.kappa, h.mass_decay) == (10, 0.8, 0.05, 100.0, 0.75, 0.99)
    # hmc.py:276: mass_collect_iters forced to 0 without adapt_mass
    assert h.mass_collect_iters == 0 and h.adapt_mass is None
    assert zs.HMC(adapt_step_size=True, adapt_mass=True).mass_collect_iters == 10


def test_latent_validation():
    h = zs.HMC()
    with pytest.raises(TypeError, match="latent\\['x'\\] is not a torch Tensor"):
        h.sample(lambda obs: obs['x'].sum(-1), {}, {'x': np.zeros((3, 2))})
    with pytest.raises(TypeError, match='float32'):
        zs.HMC().sample(lambda obs: 0, {},
                        {'x': torch.zeros(3, 2, dtype=torch.float64)})
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        zs.HMC().sample(lambda obs: 0, {}, {'x': torch.zeros(3, 2)})


def test_flags_and_placeholders():
    ph = zs.placeholder(bool, name='adapt')
    assert _flag_value(True, None, 'f') is True
    assert _flag_value(ph, {ph: 0}, 'f') is False
    assert _flag_value(ph, {ph: True}, 'f') is True
    with pytest.raises(ValueError, match='must feed a value'):
        _flag_value(ph, {}, 'f')
    assert _flag_value(zs.placeholder(bool, default=True), None, 'f') is True


def test_seeding_is_reproducible_and_distinct():
    zs.set_random_seed(1)
    a, b = zs.HMC().seed, zs.HMC().seed
    zs.set_random_seed(1)
    assert (zs.HMC().seed, zs.HMC().seed) == (a, b) and a != b
    assert zs.HMC(seed=42).seed == 42


def test_bayesian_net_structure_errors():
    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        bn.deterministic('d', torch.zeros(2))
        with pytest.raises(ValueError, match="Names should be unique"):
            bn.stochastic('s', _FakeDist())
            bn.stochastic('s', _FakeDist())
        return bn

    bn = model().observe(s=torch.zeros(2))
    with pytest.raises(TypeError, match='Expected string'):
        bn.get([3])
    with pytest.raises(ValueError, match="isn't a node named 'zz'"):
        bn.get('zz')
    with pytest.raises(ValueError, match="Node 'd' is deterministic"):
        bn.cond_log_prob('d')
    assert bn['s'].is_observed() and bn['s'].name == 's'
    assert torch.equal(bn['d'], torch.zeros(2))


class _FakeDist(object):
    dtype = torch.float32

    def _device(self):
        return None

    def get_batch_shape(self):
        return torch.Size([2])

    def get_value_shape(self):
        return torch.Size([])

    def log_prob(self, given):
        return given.sum(-1)

    def sample(self, n_samples=None):
        return torch.zeros(2)


def test_meta_bn_observe_and_log_joint_override():
    calls = []

    @zs.meta_bayesian_net()
    def model(k):
        bn = zs.BayesianNet()
        bn.stochastic('a', _FakeDist())
        bn.stochastic('b', _FakeDist())
        calls.append(k)
        return bn

    m = model(7)
    assert isinstance(m, zs.MetaBayesianNet)
    bn = m.observe(a=torch.ones(2))
    assert calls == [7]
    assert bn['a'].is_observed() and not bn['b'].is_observed()
    # default: sum of cond_log_p over stochastic nodes (bn.py:454-458)
    assert float(bn.log_joint()) == 2.0
    m.log_joint = lambda bn_: bn_.cond_log_prob('a') * 10
    assert float(m.observe(a=torch.ones(2)).log_joint()) == 20.0
    m.log_joint = 3
    with pytest.raises(TypeError, match='non-callable'):
        m.observe(a=torch.ones(2)).log_joint()
    with pytest.raises(ValueError, match='Cannot reuse'):
        zs.MetaBayesianNet(lambda: None, reuse_variables=True)


def test_observation_shape_check():
    bn = zs.BayesianNet()
    with pytest.raises(ValueError, match='Incompatible shapes'):
        zs.StochasticTensor(bn, 'x', _FakeDist(), observation=torch.zeros(3))


def test_distribution_constructor_errors():
    D = zs.distributions
    with pytest.raises(ValueError, match='Either `std` or `logstd`'):
        D.Normal(torch.zeros(2))
    with pytest.raises(ValueError, match='Either `std` or `logstd`'):
        D.Normal(torch.zeros(2), std=torch.ones(2), logstd=torch.zeros(2))
    with pytest.raises(ValueError, match='broadcastable'):
        D.Normal(torch.zeros(2), std=torch.ones(3))
    with pytest.raises(TypeError, match='float dtype'):
        D.Normal(torch.zeros(2, dtype=torch.int32), std=torch.ones(2))
    with pytest.raises(ValueError, match='non-negative'):
        D.Normal(torch.zeros(2), std=torch.ones(2), group_ndims=-1)
    with pytest.raises(ValueError, match='rank >= 1'):
        D.Categorical(torch.tensor(0.))
    n = D.Normal(torch.zeros(4, 3), logstd=torch.zeros(3), group_ndims=1)
    assert tuple(n.get_batch_shape()) == (4, 3)
    assert tuple(n.get_value_shape()) == ()
    with pytest.raises(ValueError, match='should be able to broadcast'):
        n.log_prob(torch.zeros(5))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        n.log_prob(torch.zeros(4, 3))
    c = D.Categorical(torch.zeros(2, 5))
    assert c.n_categories == 5 and tuple(c.get_batch_shape()) == (2,)
    u = D.UnnormalizedMultinomial(torch.zeros(2, 5))
    assert tuple(u.get_value_shape()) == (5,)
    with pytest.raises(NotImplementedError, match='does not support sampling'):
        u.sample()


def test_shard_bounds_cover_axis():
    for n, w in [(10, 3), (65536 * 8, 8), (5, 8), (7, 1)]:
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_merge_dicts_later_wins():
    assert zs.merge_dicts({'a': 1, 'b': 2}, {'b': 3}) == {'a': 1, 'b': 3}


def test_product_ess_matches_reference_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'ess_fixture.npz'))
    for case in ('iid', 'ar1', 'rwmh', 'sticky_f32'):
        s = fx[case + '_samples']
        for burn in (0, 100):
            np.testing.assert_allclose(
                zs.diagnostics.effective_sample_size(s, burn_in=burn),
                fx['%s_ess_burn%d' % (case, burn)], rtol=1e-10)
            batch = zs.diagnostics.effective_sample_size_batch(s, burn_in=burn)
            assert np.isclose(batch[batch > 0].min(),
                              fx['%s_ess_burn%d' % (case, burn)], rtol=1e-6)
        np.testing.assert_allclose(
            zs.diagnostics.effective_sample_size_batch(s, burn_in=0),
            fx[case + '_ess1d'], rtol=1e-6)


def test_observation_frames_nest_unwind_and_are_per_thread():
    """MetaBayesianNet.observe pushes an (owner, observed) frame for the
    duration of the builder call: nested models see their own frame, an
    exception unwinds it, other threads see none."""
    import threading
    from zhusuan_amd.framework.meta_bn import active_frame

    @zs.meta_bayesian_net()
    def inner():
        bn = zs.BayesianNet()
        bn.stochastic('z', _FakeDist())
        return bn

    seen = {}

    @zs.meta_bayesian_net()
    def outer(fail):
        bn = zs.BayesianNet()
        bn.stochastic('x', _FakeDist())
        seen['inner'] = inner().observe(z=torch.ones(2))
        seen['frame_after_inner'] = active_frame().observed
        if fail:
            raise RuntimeError('boom')
        t = threading.Thread(target=lambda: seen.update(other=active_frame()))
        t.start()
        t.join()
        return bn

    assert active_frame() is None
    bn = outer(False).observe(x=torch.zeros(2))
    assert bn['x'].is_observed() and 'z' not in bn.nodes
    assert seen['inner']['z'].is_observed()
    assert set(seen['frame_after_inner']) == {'x'}
    assert seen['other'] is None and active_frame() is None
    with pytest.raises(RuntimeError, match='boom'):
        outer(True).observe(x=torch.zeros(2))
    assert active_frame() is None
    # a BayesianNet built outside any observe() has no owner: default joint
    free = zs.BayesianNet()
    free.stochastic('a', _FakeDist())
    assert not free['a'].is_observed()


def test_placeholder_deferred_and_session_fetch_structure():
    from collections import namedtuple
    flag = zs.placeholder(bool, name='flag')
    with pytest.raises(ValueError, match='must feed'):
        flag.value
    n = zs.placeholder(int, default=3)
    assert n.value == 3
    n.feed(5)
    assert n.value == 5
    from zhusuan_amd.hmc import bind_feed
    bind_feed({n: 7, 'not a placeholder': 1})
    assert n.value == 7
    d = zs.deferred(lambda: n.value * 2)
    assert d.value == 14
    Info = namedtuple('Info', 'q alpha')
    fetched = zs.Session().run(
        [Info(q={'x': torch.ones(2)}, alpha=[torch.zeros(1), 4]), torch.ones(1)])
    assert isinstance(fetched[0], Info)
    np.testing.assert_array_equal(fetched[0].q['x'], np.ones(2))
    assert fetched[0].alpha[1] == 4 and isinstance(fetched[0].alpha[0], np.ndarray)
    single = zs.Session().run(Info(q=torch.ones(1), alpha=None))
    assert isinstance(single, Info) and single.alpha is None

    # a dict subclass whose constructor takes something else than (key, value)
    # pairs -- HMCInfo.init_momentum regenerates p0 on access -- is fetched
    # as a plain dict of arrays
    class Lazy(dict):
        def __init__(self, plan):
            super(Lazy, self).__init__()
            self._plan = plan

        def __getitem__(self, k):
            return torch.full((2,), float(self._plan[k]))

        def keys(self):
            return list(self._plan)

    got = zs.Session().run([Lazy({'x': 3, 'y': 4})])[0]
    assert type(got) is dict and sorted(got) == ['x', 'y']
    np.testing.assert_array_equal(got['y'], np.full(2, 4.0, np.float32))


def test_native_plan_reprobes_the_model_on_a_meta_latent(monkeypatch):
    """The per-run re-evaluation of the model function (fed / updated
    parameters must be seen, SURVEY section 7 "no stale caching") gets a META
    tensor for the latent: `torch.softmax(eta, -1)` in the model function then
    is shape arithmetic -- no ATen launch and no [rows, K] temporary per
    transition -- while the parameter tensors are found again by identity."""
    import torch
    import zhusuan_amd as zs
    from zhusuan_amd import hmc as H
    from zhusuan_amd.plans import recognise as R
    n_chains, n_docs, K, V = 4, 6, 8, 20
    phi = torch.softmax(torch.randn(K, V), -1)
    x = torch.poisson(torch.full((n_docs, V), 2.0))
    eta_mean = zs.placeholder(torch.float32, name='eta_mean',
                              default=torch.zeros(n_docs, K))
    eta_logstd = torch.zeros(K)
    seen = []

    @zs.meta_bayesian_net()
    def lntm():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', eta_mean.value, logstd=eta_logstd,
                        n_samples=n_chains, group_ndims=1)
        seen.append(eta.tensor.device.type)
        bn.unnormalized_multinomial(
            'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
            normalize_logits=False, dtype=torch.float32)
        return bn

    class Stub(object):
        def __init__(self, hmc, names, values, cs, dev, probe, kind):
            self.probe, self.kind = probe, kind

    monkeypatch.setattr(R, '_DenseLikelihoodPlan', Stub)
    hmc = zs.HMC(step_size=1e-3)
    hmc._observed = {'x': x}
    q = torch.zeros(n_chains, n_docs, K)
    plan = H._try_dense_likelihood_plan(hmc, lntm(), ['eta'], [q],
                                        (n_chains, n_docs), q.device)
    assert plan is not None and plan.kind == 'mixture_multinomial'
    assert seen == ['cpu']          # the build-time analysis: real latent
    del seen[:]
    ((mean, (how, spread)),), (phi_seen,), x_seen = plan.probe()
    assert seen == ['meta']
    assert mean is eta_mean.value and spread is eta_logstd and how == 'logstd'
    assert phi_seen is phi and x_seen is x
    # a newly fed prior mean is what the next probe returns
    new_mean = torch.ones(n_docs, K)
    eta_mean._value = new_mean
    assert plan.probe()[0][0][0] is new_mean and seen == ['meta', 'meta']

    # a model function that mixes the latent with device tensors cannot run
    # on a meta latent: it is evaluated on the latent itself from then on
    bias = torch.zeros(K)
    del seen[:]

    @zs.meta_bayesian_net()
    def shifted():
        bn = zs.BayesianNet()
        eta = bn.normal('eta', torch.zeros(n_docs, K), logstd=eta_logstd,
                        n_samples=n_chains, group_ndims=1)
        seen.append(eta.tensor.device.type)
        _ = eta.tensor + bias       # meta + cpu: refused by torch
        bn.unnormalized_multinomial(
            'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
            normalize_logits=False, dtype=torch.float32)
        return bn

    hmc2 = zs.HMC(step_size=1e-3)
    hmc2._observed = {'x': x}
    plan2 = H._try_dense_likelihood_plan(hmc2, shifted(), ['eta'], [q],
                                         (n_chains, n_docs), q.device)
    del seen[:]
    plan2.probe()
    assert seen == ['meta', 'cpu']
    plan2.probe()
    assert seen == ['meta', 'cpu', 'cpu']


def test_normal_derived_spread_keeps_the_tape_after_a_no_grad_first_use():
    """Normal.logstd / .std are derived lazily; a first access
    under torch.no_grad() must not cache a tensor without grad_fn."""
    import zhusuan_amd as zs
    std = torch.tensor([0.5, 2.0], requires_grad=True)
    d = zs.distributions.Normal(torch.zeros(2), std=std)
    with torch.no_grad():
        first = d.logstd
    assert not first.requires_grad
    later = d.logstd
    assert later.requires_grad and later.grad_fn is not None
    g, = torch.autograd.grad(later.sum(), std)
    torch.testing.assert_close(g, 1.0 / std.detach())
    assert d.logstd is later                 # cached from then on
    logstd = torch.tensor([0.1, -0.3], requires_grad=True)
    d2 = zs.distributions.Normal(torch.zeros(2), logstd=logstd)
    with torch.no_grad():
        d2.std
    assert d2.std.requires_grad
    # constants are cached whatever the mode
    d3 = zs.distributions.Normal(torch.zeros(2), std=torch.ones(2))
    with torch.no_grad():
        a = d3.logstd
    assert d3.logstd is a


def test_row_period_parameters_are_16_byte_aligned_copies_of_views():
    """a contiguous slice of a user tensor is a view whose storage
    offset can break the 16-byte alignment the row kernels require."""
    from zhusuan_amd.plans.dense import _to_row_period, _aligned16
    base = torch.arange(9 * 8, dtype=torch.float32).reshape(9, 8)
    view = base[:, 1:5].contiguous()[1:]          # contiguous, offset 16 B
    odd = torch.arange(33, dtype=torch.float32)[1:]   # offset 4 B
    assert odd.data_ptr() % 16 != 0
    fixed = _aligned16(odd)
    assert fixed.data_ptr() % 16 == 0 and torch.equal(fixed, odd)
    mat, rows = _to_row_period(odd[:32].reshape(4, 8), (5, 4), 8)
    assert rows == 4 and mat.data_ptr() % 16 == 0
    assert torch.equal(mat, odd[:32].reshape(4, 8))
    same, _ = _to_row_period(view, (3, 8), 4)
    assert same.data_ptr() % 16 == 0


def test_native_plan_recognises_several_latents_per_likelihood(monkeypatch):
    """VERDICT r2 item 8 on the host side: `w @ X.T + b`, two weight blocks,
    a bias given as a [C, 1] latent -- one term per latent, handed to the
    plan in the order of the latents with the user's own tensors; a latent
    used twice, a prior that depends on another latent, a user log-joint over
    two latents and more than 1 024 features are left to the generic plan."""
    import torch
    import zhusuan_amd as zs
    from zhusuan_amd import hmc as H
    from zhusuan_amd.plans import recognise as R
    C, N = 6, 30
    X1, X2 = torch.randn(N, 5), torch.randn(N, 3)
    y = (torch.rand(N) < 0.5).to(torch.float32)

    class Stub(object):
        def __init__(self, hmc, names, values, cs, dev, probe, kind):
            self.probe, self.kind, self.names = probe, kind, names

    monkeypatch.setattr(R, '_DenseLikelihoodPlan', Stub)

    def plan_of(model_fn, latents, log_joint=None):
        m = model_fn()
        if log_joint is not None:
            m.log_joint = log_joint
        hmc = zs.HMC(step_size=1e-3)
        hmc._observed = {'y': y}
        names = list(latents)
        return H._try_dense_likelihood_plan(
            hmc, m, names, [latents[k] for k in names], (C,),
            torch.device('cpu'))

    def regression(spell, use='uvbc'):
        # (only the latents in `use` exist: an unobserved node would be
        # sampled, on the device)
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            t = dict.fromkeys('uvbc')
            if 'u' in use:
                t['u'] = bn.normal('u', torch.zeros(5), std=1., n_samples=C,
                                   group_ndims=1).tensor
            if 'v' in use:
                t['v'] = bn.normal('v', torch.zeros(3), std=0.5, n_samples=C,
                                   group_ndims=1).tensor
            if 'b' in use:
                t['b'] = bn.normal('b', torch.zeros(()), std=2.,
                                   n_samples=C).tensor
            if 'c' in use:
                t['c'] = bn.normal('c', torch.zeros(1), std=2., n_samples=C,
                                   group_ndims=1).tensor
            bn.bernoulli('y', spell(t['u'], t['v'], t['b'], t['c']),
                         group_ndims=1, dtype=torch.float32)
            return bn
        return model

    q = {'u': torch.zeros(C, 5), 'v': torch.zeros(C, 3),
         'b': torch.zeros(C), 'c': torch.zeros(C, 1)}

    def pick(*names):
        return {k: q[k] for k in names}

    # weights + per-chain scalar bias; latent order != term order
    p = plan_of(regression(lambda u, v, b, c: b[:, None] + u @ X1.t(), 'ub'),
                pick('u', 'b'))
    assert p is not None and p.kind == 'linear_bernoulli'
    priors, inner, obs = p.probe()
    assert len(priors) == 2 and obs is y
    assert inner[0].data_ptr() == X1.data_ptr() and inner[1] is None
    # two weight blocks + a [C, 1] bias latent
    p = plan_of(regression(lambda u, v, b, c: u @ X1.t() + v @ X2.t() + c,
                           'uvc'), pick('u', 'v', 'c'))
    assert p is not None and p.kind == 'linear_bernoulli'
    priors, inner, obs = p.probe()
    assert [None if t is None else tuple(t.shape) for t in inner] == [
        (N, 5), (N, 3), None]
    assert priors[1][1][0] == 'std' and float(priors[1][1][1]) == 0.5
    # the explicit spelling
    p = plan_of(regression(lambda u, v, b, c: zs.linear_logits(u, X1, bias=b),
                           'ub'), pick('u', 'b'))
    assert p is not None and p.probe()[1][1] is None

    # -- refused: the generic plan samples these ------------------------------
    # a latent that enters the logits twice
    assert plan_of(regression(
        lambda u, v, b, c: u @ X1.t() + u @ X1.t(), 'u'), pick('u')) is None
    # a latent of the model that the logits do not use (its prior is a third
    # stochastic node: not "priors + one likelihood")
    assert plan_of(regression(lambda u, v, b, c: u @ X1.t() + b[:, None],
                              'ubv'), pick('u', 'b', 'v')) is None
    # a user log-joint over two latents
    assert plan_of(regression(lambda u, v, b, c: u @ X1.t() + b[:, None],
                              'ub'), pick('u', 'b'),
                   log_joint=lambda bn: bn.cond_log_prob('u') +
                   bn.cond_log_prob('b') + bn.cond_log_prob('y')) is None

    # a prior whose scale is another latent
    @zs.meta_bayesian_net()
    def hierarchical():
        bn = zs.BayesianNet()
        tau = bn.normal('b', torch.zeros(()), std=1., n_samples=C)
        u = bn.normal('u', torch.zeros(5),
                      logstd=tau.tensor[:, None] * torch.ones(5),
                      group_ndims=1)
        bn.bernoulli('y', u.tensor @ X1.t(), group_ndims=1,
                     dtype=torch.float32)
        return bn
    assert plan_of(hierarchical, pick('u', 'b')) is None

    # more than 1 024 features in total
    Xw = torch.randn(N, 1100)

    @zs.meta_bayesian_net()
    def too_wide():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(1100), std=1., n_samples=C,
                      group_ndims=1)
        bn.bernoulli('y', w.tensor @ Xw.t(), group_ndims=1,
                     dtype=torch.float32)
        return bn
    assert plan_of(too_wide, {'w': torch.zeros(C, 1100)}) is None


def test_plan_recognition_survives_a_foreign_autograd_function(monkeypatch):
    """a MetaBayesianNet model that passes a latent
    through a user torch.autograd.Function.  sample()'s first evaluation runs
    on latents without grad, so no SymbolicCut fires there; the plan
    recognisers then re-run the model with requires_grad symbols and the cut
    fires inside THEM -- it must not escape HMC.sample(): plain tensors from
    then on, generic plan."""
    import torch
    import zhusuan_amd as zs
    from zhusuan_amd import _symbolic, hmc as H
    from zhusuan_amd.plans import recognise as R
    C, D, N = 5, 4, 12
    X = torch.randn(N, D)
    y = (torch.rand(N) < 0.5).to(torch.float32)

    class Scale(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.5

        @staticmethod
        def backward(ctx, g):
            return g * 1.5

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D), std=1., n_samples=C,
                      group_ndims=1)
        bn.bernoulli('y', Scale.apply(w.tensor) @ X.t(), group_ndims=1,
                     dtype=torch.float32)
        return bn

    class Stub(object):
        def __init__(self, *a):
            raise AssertionError('a native plan was built')

    monkeypatch.setattr(R, '_DenseLikelihoodPlan', Stub)
    q = torch.zeros(C, D)
    hmc = zs.HMC(step_size=1e-3)
    hmc._observed = {'y': y}
    # the recogniser alone raises (what used to escape) ...
    with pytest.raises(_symbolic.SymbolicCut):
        H._try_dense_likelihood_plan(hmc, model(), ['w'], [q], (C,), q.device)
    # ... the sampler's wrapper turns it into the generic plan
    assert hmc._symbolic_latents is True
    assert hmc._recognise_plan(model(), ['w'], [q], (C,), q.device) is None
    assert hmc._symbolic_latents is False


def test_native_plan_recognises_the_softmax_regression_spellings(monkeypatch):
    """Row J1 of VERDICT r3 on the host side: y ~ Categorical(X @ w^T), w a
    [K, F] latent per chain with a group_ndims = 2 Normal prior -- the
    literal `X.unsqueeze(0) @ w.transpose(-1, -2)`, `X @ w.mT`, the
    transposed `w @ X.T` and zs.linear_class_logits all reach the
    'linear_categorical' plan with the user's own tensors; near misses and
    over-wide class counts are refused with a reason, aloud."""
    import warnings
    import torch
    import zhusuan_amd as zs
    from zhusuan_amd import hmc as H
    from zhusuan_amd.plans import recognise as R
    C, K, F, N = 6, 4, 5, 30
    X = torch.randn(N, F)
    y = torch.randint(0, K, (N,), dtype=torch.int32)

    class Stub(object):
        def __init__(self, hmc, names, values, cs, dev, probe, kind):
            self.probe, self.kind = probe, kind

    monkeypatch.setattr(R, '_DenseLikelihoodPlan', Stub)

    def plan_of(spell, n_classes=K):
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            w = bn.normal('w', torch.zeros(n_classes, F), std=1.,
                          n_samples=C, group_ndims=2)
            bn.categorical('y', spell(w.tensor), group_ndims=1)
            return bn
        hmc = zs.HMC(step_size=1e-3)
        hmc._observed = {'y': y}
        q = torch.zeros(C, n_classes, F)
        return hmc, H._try_dense_likelihood_plan(
            hmc, model(), ['w'], [q], (C,), torch.device('cpu'))

    for spell in (lambda w: X.unsqueeze(0) @ w.transpose(-1, -2),
                  lambda w: X @ w.mT,
                  lambda w: (w @ X.t()).transpose(1, 2),
                  lambda w: X.unsqueeze(0).expand(C, N, F) @ w.transpose(1, 2),
                  lambda w: zs.linear_class_logits(w, X)):
        hmc, p = plan_of(spell)
        assert p is not None and p.kind == 'linear_categorical'
        (prior,), (X_seen,), y_seen = p.probe()
        assert X_seen.data_ptr() == X.data_ptr() and y_seen is y
        assert prior[1][0] == 'std'
    # a near miss materialises the logits: refused, and said so
    hmc, p = plan_of(lambda w: X.unsqueeze(0) @ (w * 1.0).transpose(-1, -2))
    assert p is None and 'materialised' in hmc._refusal[0]
    # a tiled copy of X carries data along the chain axis: not recognised
    hmc, p = plan_of(lambda w: X.unsqueeze(0).repeat(C, 1, 1) @
                     w.transpose(1, 2))
    assert p is None
    # 40 classes: the lazy likelihood is there, the kernel's limit is not
    hmc, p = plan_of(lambda w: X.unsqueeze(0) @ w.transpose(-1, -2), 40)
    assert p is None and '40' in hmc._refusal[0]


def test_csr_segments_cover_every_row_once():
    """_ops._csr_segments (the pair list of zshmc_gather_dot_normal_lik_grad):
    every latent row has at least one segment -- rows without pairs an empty
    one --, no segment is longer than GD_SEGMENT_PAIRS, consecutive segments
    are consecutive CSR slots, and a row's segments are contiguous."""
    import torch
    from zhusuan_amd import _ops
    rng = np.random.RandomState(0)
    n_rows = 37
    counts = rng.poisson(120, size=n_rows)
    counts[[3, 9]] = 0
    counts[5] = 700                       # three segments
    counts[20] = _ops.GD_SEGMENT_PAIRS    # exactly one full segment
    seg = torch.zeros(n_rows + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(torch.tensor(counts), 0).to(torch.int32)
    E = int(counts.sum())
    sp, sr, sf, lr = _ops._csr_segments(seg, E)
    sp, sr, sf, lr = (t.numpy() for t in (sp, sr, sf, lr))
    assert sp[0] == 0 and sp[-1] == E and (np.diff(sp) >= 0).all()
    assert np.diff(sp).max() <= _ops.GD_SEGMENT_PAIRS
    assert (np.diff(sr) >= 0).all() and set(sr) == set(range(n_rows))
    for i in range(n_rows):
        mine = np.nonzero(sr == i)[0]
        assert mine[0] == sf[i] and (np.diff(mine) == 1).all()
        assert sp[mine[0]] == seg[i] and sp[mine[-1] + 1] == seg[i + 1]
    assert list(lr) == [5]
    assert len(sr) == n_rows + 2


def test_row_range_slices_policy(monkeypatch):
    """_ops._row_splits: no slices once the chain blocks fill the device's
    resident slots; otherwise about two workgroups per CU, slices of at least
    two 64-row tiles, at most 256; a kernel that holds two workgroups per CU
    gets its partner from slices when the chain blocks are one wave."""
    import types
    import torch
    from zhusuan_amd import _ops
    monkeypatch.setattr(torch.cuda, 'get_device_properties',
                        lambda d: types.SimpleNamespace(multi_processor_count=256))
    f = _ops._row_splits
    assert f(32768, 10 ** 6, None, 64) == 1              # 512 blocks
    assert f(100, 12419, None, 64) == 98                 # the E-step: 2 blocks
    assert f(100, 200, None, 64) == 2                    # short inner range
    assert f(64 * 300, 10 ** 6, None, 64) == 1
    assert f(128 * 256, 10 ** 6, None, 128, per_cu=2) == 2   # bf16x3, 128 columns
    assert f(128 * 256, 10 ** 6, None, 128, per_cu=1) == 1
    assert f(64, 10 ** 6, None, 64) == 256
    assert _ops.resident_per_cu(128, 'bf16x3') == 2
    assert _ops.resident_per_cu(256, 'bf16x3') == 1
    assert _ops.resident_per_cu(64) == 3 and _ops.resident_per_cu(512) == 1


def test_forget_drops_one_model_s_cached_operands_only():
    """reuse_start_evaluation=False used to clear the process-wide
    operand caches on every run; it now drops what was built from ITS model's
    tensors (and what was built from those in turn), nothing else."""
    import torch
    from zhusuan_amd import _ops
    _ops.clear_caches()
    Xa, Xb = torch.randn(6, 5), torch.randn(7, 5)
    pa, pb = _ops._padded_x(Xa, 8), _ops._padded_x(Xb, 8)
    assert _ops._padded_x(Xa, 8) is pa and _ops._padded_x(Xb, 8) is pb
    # something derived from the padded copy of A (as the bf16x3 image is)
    _ops._image_cache.put(_ops._tensor_key(pa), 'image of A', pa)
    _ops._image_cache.put(_ops._tensor_key(pb), 'image of B', pb)
    phi = torch.rand(3, 9)
    pt = _ops._padded_phi_t(phi, 4)
    _ops.forget([Xa.t().t()])                    # a view: same storage
    assert _ops._padded_x(Xb, 8) is pb
    assert _ops._image_cache.get(_ops._tensor_key(pb)) == 'image of B'
    assert _ops._image_cache.get(_ops._tensor_key(pa)) is None
    assert _ops._padded_phi_t(phi, 4) is pt
    assert _ops._padded_x(Xa, 8) is not pa
    _ops.forget([phi, None])
    assert _ops._padded_phi_t(phi, 4) is not pt
    _ops.clear_caches()


def test_write_generations_never_repeat():
    """generations come from ONE counter, so an entry that was
    evicted and made again cannot equal a value recorded before."""
    import torch
    from zhusuan_amd import _writes
    a, b = torch.zeros(3), torch.zeros(3)
    _writes.note([a])
    ga = _writes.generation(a)
    _writes.note([b])
    assert _writes.generation(b) > ga
    _writes._generation.pop(_writes._key(a))     # "evicted"
    assert _writes.generation(a) == 0
    _writes.note([a])
    assert _writes.generation(a) > _writes.generation(b) > ga
    # ... and it is part of every operand-cache key
    from zhusuan_amd import _ops
    k0 = _ops._tensor_key(a)
    _writes.note([a])
    assert _ops._tensor_key(a) != k0


def test_counts_csr_pads_every_document_to_whole_tiles():
    """_ops.counts_csr (the documents' OWN vocabularies of the ABI 0.6.0
    topic-model kernels): compacted counts, the words' rows, offsets; every
    document's slice a multiple of 32, at least 32, padded with count 0 /
    row 0; row-major order inside a document; cached per tensor version."""
    import numpy as np
    import torch
    from zhusuan_amd import _ops
    _ops.clear_caches()
    rng = np.random.RandomState(0)
    x = rng.poisson(0.1, size=(7, 300)).astype(np.float32)
    x[2] = 0.0                                   # a document without words
    x[4] = 1.0 + rng.poisson(1.0, size=300)      # one that uses every word
    xt = torch.tensor(x)
    vals, rows, off, total = _ops.counts_csr(xt)
    off = off.numpy()
    assert off[0] == 0 and off[-1] == total == vals.numel() == rows.numel()
    lens = np.diff(off)
    assert (lens % 32 == 0).all() and (lens >= 32).all()
    assert lens[2] == 32 and lens[4] == 320
    for d in range(7):
        v = vals[off[d]:off[d + 1]].numpy()
        r = rows[off[d]:off[d + 1]].numpy()
        nz = np.flatnonzero(x[d])
        np.testing.assert_array_equal(r[:len(nz)], nz)
        np.testing.assert_array_equal(v[:len(nz)], x[d][nz])
        assert (v[len(nz):] == 0).all() and (r[len(nz):] == 0).all()
    assert _ops.counts_csr(xt)[0] is vals        # cached
    xt[0, 0] += 1.0                              # a torch write: new version
    assert _ops.counts_csr(xt)[0] is not vals
    _ops.clear_caches()
