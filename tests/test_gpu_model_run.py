"""sample_op.run_many / sample_op.anneal on the NATIVE model plans: the launch
loop of n transitions inside libzshmc.so (zshmc_hmc_model_run,
csrc/hmc_model_run.hip) against n single `sample_op.run` calls from Python --
bit-identical latents, HMCInfo, step size, tuner state, EWMV state and mass,
for every plan kind (dense-logit Bernoulli with one and with three latents,
the topic model's mixture multinomial, dense-logit Categorical, the
gathered-dot rating model), with the step size / mass adapting inside the
block, held, or absent; and AIS's annealing loop (evaluation.py:119-165) from
one call against its Python loop.  Reference: zhusuan/hmc.py:418-520 (one
`sample_op` execution), examples/topic_models/lntm_mcem.py:157-182 (the
E-step loop that motivates it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _blr(zs, torch, dev, C=96):
    g = torch.Generator(device=dev).manual_seed(1)
    N, D = 300, 20
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.5).float()
    zero = torch.zeros(D, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', zero, std=1., n_samples=C, group_ndims=1)
        bn.bernoulli('y', w.tensor @ X.t(), group_ndims=1, dtype=torch.float32)
        return bn
    return model, {'y': y}, lambda: {'w': torch.zeros(C, D, device=dev)}, \
        'linear_bernoulli'


def _blr3(zs, torch, dev, C=64):
    g = torch.Generator(device=dev).manual_seed(2)
    N = 200
    X1 = torch.randn(N, 7, device=dev, generator=g)
    X2 = torch.randn(N, 6, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.5).float()
    z7, z6, z0 = (torch.zeros(7, device=dev), torch.zeros(6, device=dev),
                  torch.zeros((), device=dev))

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        u = bn.normal('u', z7, std=1., n_samples=C, group_ndims=1)
        v = bn.normal('v', z6, std=0.5, n_samples=C, group_ndims=1)
        b = bn.normal('b', z0, std=2., n_samples=C)
        bn.bernoulli('y', u.tensor @ X1.t() + v.tensor @ X2.t() +
                     b.tensor.unsqueeze(1), group_ndims=1,
                     dtype=torch.float32)
        return bn
    return model, {'y': y}, lambda: {
        'u': torch.zeros(C, 7, device=dev), 'v': torch.zeros(C, 6, device=dev),
        'b': torch.zeros(C, device=dev)}, 'linear_bernoulli'


def _lntm(zs, torch, dev, n_chains=8, n_docs=12, K=10, V=60):
    g = torch.Generator(device=dev).manual_seed(3)
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 1.5, device=dev), generator=g)
    mean = 0.1 * torch.randn(n_docs, K, device=dev, generator=g)
    logstd = torch.zeros(K, device=dev)

    def build():
        @zs.meta_bayesian_net()
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', mean, logstd=logstd, n_samples=n_chains,
                            group_ndims=1)
            theta = torch.softmax(eta.tensor, -1)
            bn.unnormalized_multinomial(
                'x', torch.log((theta.reshape(-1, K) @ phi).reshape(
                    n_chains, n_docs, V)), normalize_logits=False,
                dtype=torch.float32)
            return bn
        m = lntm()
        m.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
        return m
    return build, {'x': x}, lambda: {
        'eta': torch.zeros(n_chains, n_docs, K, device=dev)}, \
        'mixture_multinomial'


def _softmax(zs, torch, dev, C=40, K=5, F=12, N=150):
    g = torch.Generator(device=dev).manual_seed(4)
    X = torch.randn(N, F, device=dev, generator=g)
    y = torch.randint(0, K, (N,), device=dev, generator=g, dtype=torch.int32)
    zero = torch.zeros(K, F, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', zero, std=1., n_samples=C, group_ndims=2)
        bn.categorical('y', X.unsqueeze(0) @ w.tensor.transpose(-1, -2),
                       group_ndims=1)
        return bn
    return model, {'y': y}, lambda: {'w': torch.zeros(C, K, F, device=dev)}, \
        'linear_categorical'


def _pmf(zs, torch, dev, K=6, n=14, m=10, D=6, E=120):
    g = torch.Generator(device=dev).manual_seed(5)
    su = torch.randint(0, n, (E,), device=dev, generator=g, dtype=torch.int32)
    sv = torch.randint(0, m, (E,), device=dev, generator=g, dtype=torch.int32)
    r = torch.rand(E, device=dev, generator=g)
    v = 0.5 * torch.randn(K, m, D, device=dev, generator=g)
    zu, zv = torch.zeros(n, D, device=dev), torch.zeros(m, D, device=dev)

    def build():
        @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
        def pmf():
            bn = zs.BayesianNet()
            u = bn.normal('u', zu, std=1.0, n_samples=K, group_ndims=1)
            vv = bn.normal('v', zv, std=1.0, n_samples=K, group_ndims=1)
            bn.normal('r', torch.sigmoid(zs.gathered_dot(u, su, vv, sv)),
                      std=0.2)
            return bn
        mm = pmf()
        mm.log_joint = lambda bn: (
            bn.cond_log_prob('u').sum(-1) + bn.cond_log_prob('v').sum(-1) +
            bn.cond_log_prob('r').sum(-1))
        return mm
    return build, {'r': r, 'v': v}, lambda: {
        'u': 0.05 * torch.ones(K, n, D, device=dev)}, 'gathered_dot'


MODELS = {'blr': _blr, 'blr3': _blr3, 'lntm': _lntm, 'softmax': _softmax,
          'pmf': _pmf}


@pytest.mark.parametrize('which', sorted(MODELS))
@pytest.mark.parametrize('adaptive', ['none', 'step', 'step+mass'])
def test_run_many_equals_a_loop_of_runs(env, which, adaptive):
    zs, torch, dev = env
    from zhusuan_amd import _capi
    build, observed, latents, kind = MODELS[which](zs, torch, dev)
    out = []
    for many in (False, True):
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        kw = dict(step_size=0.02, n_leapfrogs=4, seed=8)
        if adaptive != 'none':
            kw.update(adapt_step_size=f_ss, target_acceptance_rate=0.7)
        if adaptive == 'step+mass':
            kw.update(adapt_mass=f_m, mass_collect_iters=3)
        hmc = zs.HMC(**kw)
        q = latents()
        op, info = hmc.sample(build(), observed, q)
        assert hmc.plan_kind == kind, hmc.plan_reason
        calls = []
        real = _capi.call

        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        _capi.call = spy
        try:
            # adaptation on (search at t = 1, re-initialisation at
            # t = mass_collect_iters), then held, then on again
            for n, feed in ((7, {f_ss: True, f_m: True}),
                            (6, {f_ss: False, f_m: False}),
                            (5, {f_ss: True, f_m: True})):
                if many:
                    op.run_many(n, feed_dict=feed)
                else:
                    for _ in range(n):
                        op.run(feed_dict=feed)
        finally:
            _capi.call = real
        st = hmc.get_state()
        out.append(dict(
            q={k: v.clone() for k, v in q.items()},
            info={f: getattr(info, f).clone() for f in (
                'acceptance_rate', 'orig_hamiltonian', 'hamiltonian',
                'orig_log_prob', 'log_prob')},
            state=st, calls=calls, t=hmc.t))
    a, b = out
    assert a['t'] == b['t'] == 18
    assert b['calls'].count('zshmc_hmc_model_run') >= 2
    # the block replaces the per-transition calls of the transitions it
    # covers (a single run is itself ONE call for the transition --
    # zshmc_hmc_model_transition -- plus its statistics and update)
    assert len(b['calls']) < len(a['calls'])
    assert a['calls'].count('zshmc_hmc_model_transition') >= 10
    for k in a['q']:
        assert torch.equal(a['q'][k], b['q'][k]), k
    for f in a['info']:
        assert torch.equal(a['info'][f], b['info'][f]), f
    assert torch.equal(a['state']['state'], b['state']['state'])
    if adaptive == 'step+mass':
        for key in ('ewmv_mean', 'ewmv_var', 'mass'):
            for x, y in zip(a['state'][key], b['state'][key]):
                assert torch.equal(x, y), key


def test_annealing_from_one_call_equals_the_python_loop(env):
    """AIS on the topic model (lntm_mcem.py:116-141): the annealing loop and
    its weight accumulation through sample_op.anneal against the reference
    loop of single runs -- the same log-weights, bit for bit."""
    zs, torch, dev = env
    from zhusuan_amd import hmc as H
    n_chains, n_docs, K, V = 6, 9, 8, 40
    g = torch.Generator(device=dev).manual_seed(6)
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 2.0, device=dev), generator=g)
    mean, logstd = torch.zeros(n_docs, K, device=dev), torch.zeros(K,
                                                                   device=dev)

    def models():
        import copy

        @zs.meta_bayesian_net()
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', mean, logstd=logstd, n_samples=n_chains,
                            group_ndims=1)
            bn.unnormalized_multinomial(
                'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
                normalize_logits=False, dtype=torch.float32)
            return bn
        target = lntm()
        target.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                       bn.cond_log_prob('x'))
        # lntm_mcem.py:128-134: the same model with the prior as log-joint
        proposal = copy.copy(target)
        proposal.log_joint = lambda bn: bn.cond_log_prob('eta')
        return target, proposal

    res = []
    for block in (True, False):
        zs.set_random_seed(123)
        target, proposal = models()
        flag = zs.placeholder(bool, default=False)
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=3, adapt_step_size=flag,
                     target_acceptance_rate=0.6, seed=77)
        eta = torch.zeros(n_chains, n_docs, K, device=dev)
        ais = zs.AIS(target, proposal, hmc, {'x': x}, {'eta': eta},
                     n_temperatures=40, n_adapt=5)
        assert hmc.plan_kind == 'mixture_multinomial'
        if not block:
            ais.verbose = True           # the Python loop (prints per step)
            import builtins
            real_print = builtins.print
            builtins.print = lambda *a, **k: None
        try:
            val = ais.run(feed_dict={flag: False})
        finally:
            if not block:
                builtins.print = real_print
        res.append((val, ais.log_weights.clone(), eta.clone()))
    assert torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])
    assert res[0][0] == res[1][0]


LIK_CALLS = ('zshmc_linear_bernoulli_log_lik', 'zshmc_linear_multinomial_log_lik',
             'zshmc_linear_categorical_log_lik', 'zshmc_gather_dot_normal_lik')


@pytest.mark.parametrize('which', sorted(MODELS))
def test_start_evaluation_is_carried_over(env, which):
    """A transition's first likelihood evaluation is the previous transition's
    last one where the chain accepted (zshmc_model_plan.grad_start / ll_start;
    SURVEY 8d: "old log-prob carried over on the model-unchanged fast path"):
    L launches per transition instead of L + 1, results bit-identical to
    evaluating every start point -- through the Python loop and through
    zshmc_hmc_model_run; an outside write to a latent makes the next
    transition evaluate again."""
    zs, torch, dev = env
    from zhusuan_amd import _capi
    build, observed, latents, kind = MODELS[which](zs, torch, dev)
    L, out = 4, {}
    for mode in ('carry', 'evaluate', 'carry_block', 'evaluate_block'):
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=L, seed=8, adapt_step_size=f_ss,
                     adapt_mass=f_m, mass_collect_iters=3,
                     target_acceptance_rate=0.7)
        q = latents()
        op, info = hmc.sample(build(), observed, q)
        assert hmc.plan_kind == kind, hmc.plan_reason
        hmc._plan.carry_start = mode.startswith('carry')
        # (count the launches of the Python loop, not one C call per transition)
        hmc._plan.c_transition = False
        calls, real = [], _capi.call

        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        feeds = ((6, {f_ss: True, f_m: True}), (5, {f_ss: False, f_m: False}))
        n_lik = []
        for n, feed in feeds:
            _capi.call = spy
            try:
                if mode.endswith('block'):
                    op.run_many(n, feed_dict=feed)
                else:
                    for _ in range(n):
                        op.run(feed_dict=feed)
            finally:
                _capi.call = real
            n_lik.append(sum(c in LIK_CALLS for c in calls))
            del calls[:]
        if not mode.endswith('block'):
            # the held stretch: no search, no first evaluation
            assert n_lik[1] == 5 * (L if mode == 'carry' else L + 1), n_lik
        # somebody else writes a latent: the start is evaluated again
        name0 = sorted(q)[0]
        q[name0].mul_(1.0)
        _capi.call = spy
        try:
            op.run(feed_dict=feeds[1][1])
        finally:
            _capi.call = real
        assert sum(c in LIK_CALLS for c in calls) == L + 1, calls
        # ... and so does an in-place change of an observed tensor (another
        # likelihood); the run after that carries again
        for expect in (L + 1, L if mode.startswith('carry') else L + 1):
            if expect == L + 1:
                observed[sorted(observed)[0]].add_(0)
            del calls[:]
            _capi.call = spy
            try:
                op.run(feed_dict=feeds[1][1])
            finally:
                _capi.call = real
            assert sum(c in LIK_CALLS for c in calls) == expect, (expect, calls)
        # a write torch's version counters do not see (x.data): announced
        q[name0].data.mul_(1.0)
        hmc.latents_changed()
        del calls[:]
        _capi.call = spy
        try:
            op.run(feed_dict=feeds[1][1])
        finally:
            _capi.call = real
        assert sum(c in LIK_CALLS for c in calls) == L + 1, calls
        out[mode] = dict(
            q={k: v.clone() for k, v in q.items()},
            info={f: getattr(info, f).clone() for f in (
                'acceptance_rate', 'orig_hamiltonian', 'hamiltonian',
                'orig_log_prob', 'log_prob')},
            state=hmc.get_state())
    ref = out['evaluate']
    for mode in ('carry', 'carry_block', 'evaluate_block'):
        got = out[mode]
        for k in ref['q']:
            assert torch.equal(ref['q'][k], got['q'][k]), (mode, k)
        for f in ref['info']:
            assert torch.equal(ref['info'][f], got['info'][f]), (mode, f)
        assert torch.equal(ref['state']['state'], got['state']['state']), mode
        for key in ('ewmv_mean', 'ewmv_var', 'mass'):
            for x, y in zip(ref['state'][key], got['state'][key]):
                assert torch.equal(x, y), (mode, key)


# -- the one-launch trajectory (csrc/hmc_model_traj.hip) -----------------------
def _estep(zs, torch, dev, n_chains=1, n_docs=100, K=100, V=1300):
    """The E-step of examples/topic_models/lntm_mcem.py:62-70,157-182 at its
    minibatch shape (one chain, 100 documents, K = 100; a vocabulary long
    enough for several row-range slices)."""
    return _lntm(zs, torch, dev, n_chains=n_chains, n_docs=n_docs, K=K, V=V)


def _blr_long(zs, torch, dev):
    """Few chains, many data rows: chain blocks x slices on one launch."""
    g = torch.Generator(device=dev).manual_seed(11)
    N, D, C = 5000, 200, 70
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.5).float()
    zero = torch.zeros(D, device=dev)

    @zs.meta_bayesian_net()
    def model():
        bn = zs.BayesianNet()
        w = bn.normal('w', zero, std=1., n_samples=C, group_ndims=1)
        bn.bernoulli('y', w.tensor @ X.t(), group_ndims=1, dtype=torch.float32)
        return bn
    return model, {'y': y}, lambda: {'w': torch.zeros(C, D, device=dev)}, \
        'linear_bernoulli'


TRAJ_MODELS = {'estep': _estep, 'blr_long': _blr_long, 'blr': _blr,
               'blr3': _blr3, 'lntm': _lntm}


@pytest.mark.parametrize('which', sorted(TRAJ_MODELS))
def test_one_launch_trajectory_is_bit_identical_to_a_launch_per_trip(env, which):
    """HMC(one_launch_trajectory=True): the L + 1 trips from one cooperative
    launch with grid barriers against one launch per likelihood evaluation /
    step -- the same device code in the same order (csrc/lb_body.h,
    csrc/model_step.h), so every latent, HMCInfo field, the step size and the
    mass are identical BIT FOR BIT; through sample_op.run and run_many, with
    the step size and the mass adapting, held, and with the start evaluation
    carried or not."""
    zs, torch, dev = env
    from zhusuan_amd import _capi
    build, observed, latents, kind = TRAJ_MODELS[which](zs, torch, dev)
    out = {}
    for one, reuse, block in ((True, True, False), (False, True, False),
                              (True, True, True), (True, False, False),
                              (False, False, True)):
        f_ss, f_m = zs.placeholder(bool), zs.placeholder(bool)
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=5, seed=21,
                     adapt_step_size=f_ss, adapt_mass=f_m, mass_collect_iters=3,
                     target_acceptance_rate=0.7, one_launch_trajectory=one,
                     reuse_start_evaluation=reuse)
        q = latents()
        op, info = hmc.sample(build(), observed, q)
        assert hmc.plan_kind == kind, hmc.plan_reason
        calls, real = [], _capi.call

        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        _capi.call = spy
        try:
            for n, feed in ((6, {f_ss: True, f_m: True}),
                            (4, {f_ss: False, f_m: False})):
                if block:
                    op.run_many(n, feed_dict=feed)
                else:
                    for _ in range(n):
                        op.run(feed_dict=feed)
        finally:
            _capi.call = real
        hmc.check_numerics()
        assert bool(hmc._plan.traj_capacity > 0) == one
        st = hmc.get_state()
        out[(one, reuse, block)] = (
            {k: v.cpu().numpy() for k, v in q.items()},
            {f: getattr(info, f).cpu().numpy() for f in (
                'acceptance_rate', 'orig_hamiltonian', 'hamiltonian',
                'orig_log_prob', 'log_prob')},
            float(info.updated_step_size.item()),
            [m.cpu().numpy() for m in hmc._plan.mass], st)
    ref = out[(False, True, False)]
    for key, got in out.items():
        for k in ref[0]:
            np.testing.assert_array_equal(got[0][k], ref[0][k], err_msg=str(key))
        for f in ref[1]:
            np.testing.assert_array_equal(got[1][f], ref[1][f],
                                          err_msg='%s %s' % (key, f))
        assert got[2] == ref[2], key
        for a, b in zip(got[3], ref[3]):
            np.testing.assert_array_equal(a, b)
    a = float(ref[1]['acceptance_rate'].mean())
    assert 0.05 < a <= 1.0, a


def test_estep_transition_time(env):
    """The E-step shape of lntm_mcem.py (100 documents x K = 100 x V = 12 419,
    L = 20): wall time per transition with and without the one-launch
    trajectory (printed: profiles/r05k_* has the numbers and why the grid
    barriers lose to kernel boundaries on this chip)."""
    import time
    zs, torch, dev = env
    build, observed, latents, kind = _lntm(zs, torch, dev, n_chains=1,
                                           n_docs=100, K=100, V=12419)
    ms = {}
    for one in (False, True):
        hmc = zs.HMC(step_size=0.05, n_leapfrogs=20, seed=5,
                     one_launch_trajectory=one)
        op, info = hmc.sample(build(), observed, latents())
        op.run_many(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        op.run_many(40)
        torch.cuda.synchronize()
        ms[one] = (time.perf_counter() - t0) / 40 * 1e3
        hmc.check_numerics()
    print('E-step transition: launch per trip %.3f ms, one launch %.3f ms' % (
        ms[False], ms[True]))
    assert ms[False] < 1.0          # (1.06 ms with 32 slices and the serial reduce)
