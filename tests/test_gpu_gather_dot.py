"""GPU parity of zs.gathered_dot (csrc/gather_dot.hip) -- the rating logits of
the reference's pmf_hmc.py:26-28 -- against the oracle restatement and float64
torch autograd, and the PMF model sampled by HMC with the fused op vs the
gather-materialising model."""
import numpy as np
import pytest

from oracle import hmc_ref, pmf_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


@pytest.mark.parametrize('lead,n,m,D,E', [((1,), 5, 7, 1, 9), ((8,), 50, 311, 30, 4000),
                                          ((3,), 64, 64, 16, 1), ((2, 3), 17, 9, 33, 500),
                                          ((4,), 10, 2000, 100, 3000)])
def test_forward_backward_match_float64(env, lead, n, m, D, E):
    zs, torch, dev = env
    rng = np.random.RandomState(E + D)
    u = rng.normal(size=lead + (n, D)).astype(np.float32)
    v = rng.normal(size=lead + (m, D)).astype(np.float32)
    su = rng.randint(0, n, size=E)
    sv = rng.randint(0, m, size=E)
    if n > 3:
        su[su == 2] = 3                       # row 2 of u receives nothing
    w = rng.normal(size=lead + (E,)).astype(np.float32)
    ut = torch.tensor(u, device=dev, requires_grad=True)
    vt = torch.tensor(v, device=dev, requires_grad=True)
    sut = torch.tensor(su, device=dev, dtype=torch.int32)
    svt = torch.tensor(sv, device=dev, dtype=torch.int64)
    out = zs.gathered_dot(ut, sut, vt, svt)
    assert tuple(out.shape) == lead + (E,)
    want = pmf_ref.gathered_dot(u.astype(np.float64), su, v.astype(np.float64), sv)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5,
                               atol=1e-5 * np.sqrt(D))
    np.testing.assert_allclose(out.detach().cpu().numpy(),
                               pmf_ref.gathered_dot(u, su, v, sv), rtol=1e-5,
                               atol=2e-6 * D)
    (out * torch.tensor(w, device=dev)).sum().backward()
    gu, gv = pmf_ref.gathered_dot_grads(u.astype(np.float64), su,
                                        v.astype(np.float64), sv,
                                        w.astype(np.float64))
    tol = dict(rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(gu).max())))
    np.testing.assert_allclose(ut.grad.cpu().numpy(), gu, **tol)
    np.testing.assert_allclose(vt.grad.cpu().numpy(), gv, **tol)
    if n > 3:
        assert not ut.grad[..., 2, :].any()
    # bit-reproducible (segmented sums, no atomics)
    g1 = ut.grad.clone()
    ut.grad = None
    (zs.gathered_dot(ut, sut, vt, svt) * torch.tensor(w, device=dev)).sum().backward()
    assert torch.equal(g1, ut.grad)


def test_argument_checks(env):
    zs, torch, dev = env
    u = torch.zeros(2, 4, 3, device=dev)
    v = torch.zeros(2, 5, 3, device=dev)
    i = lambda a: torch.tensor(a, device=dev, dtype=torch.int32)
    with pytest.raises(IndexError):
        zs.gathered_dot(u, i([0, 4]), v, i([0, 0]))
    with pytest.raises(IndexError):
        zs.gathered_dot(u, i([0, 1]), v, i([-1, 0]))
    with pytest.raises(ValueError):
        zs.gathered_dot(u, i([0, 1]), v, i([0]))
    with pytest.raises(ValueError):
        zs.gathered_dot(u, i([0]), torch.zeros(3, 5, 3, device=dev), i([0]))
    with pytest.raises(TypeError):
        zs.gathered_dot(u, torch.zeros(2, device=dev), v, i([0, 0]))
    with pytest.raises(RuntimeError):
        zs.gathered_dot(u.cpu(), i([0]), v, i([0]))
    # an index tensor edited in place is re-validated / re-sorted
    idx = i([0, 1, 2])
    a = zs.gathered_dot(torch.ones(1, 4, 2, device=dev) *
                        torch.arange(4, device=dev).view(1, 4, 1), idx,
                        torch.ones(1, 5, 2, device=dev), i([0, 0, 0]))
    np.testing.assert_allclose(a.cpu().numpy(), [[0, 2, 4]])
    idx[0] = 3
    a = zs.gathered_dot(torch.ones(1, 4, 2, device=dev) *
                        torch.arange(4, device=dev).view(1, 4, 1), idx,
                        torch.ones(1, 5, 2, device=dev), i([0, 0, 0]))
    np.testing.assert_allclose(a.cpu().numpy(), [[6, 2, 4]])


def _pmf_problem(rng, K, n, m, D, E):
    su = np.sort(rng.randint(0, n, size=E))
    sv = rng.randint(0, m, size=E)
    u_true = rng.normal(size=(n, D)) * 0.6
    v_true = rng.normal(size=(m, D)) * 0.6
    r = 1 / (1 + np.exp(-(u_true[su] * v_true[sv]).sum(-1)))
    r = (r + 0.05 * rng.normal(size=E)).astype(np.float32)
    u0 = (0.1 * rng.normal(size=(K, n, D))).astype(np.float32)
    v = (v_true + 0.05 * rng.normal(size=(K, m, D))).astype(np.float32)
    return su, sv, r, u0, v


def test_pmf_hmc_fused_vs_dense_vs_oracle(env):
    """pmf_hmc.py:19-31,121-147: HMC over u given v and the ratings."""
    zs, torch, dev = env
    rng = np.random.RandomState(2)
    K, n, m, D, E = 8, 50, 120, 30, 1500
    su, sv, r, u0, v = _pmf_problem(rng, K, n, m, D, E)
    alpha_u = alpha_v = 1.0
    alpha_pred = 0.2 / 4.0
    T = lambda a, **kw: torch.tensor(a, device=dev, **kw)
    sut, svt, rt, vt = T(su, dtype=torch.int32), T(sv, dtype=torch.int32), T(r), T(v)

    def build(fused):
        @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
        def pmf():
            bn = zs.BayesianNet()
            u = bn.normal('u', torch.zeros(n, D, device=dev), std=alpha_u,
                          n_samples=K, group_ndims=1)
            vv = bn.normal('v', torch.zeros(m, D, device=dev), std=alpha_v,
                           n_samples=K, group_ndims=1)
            if fused:
                r_logits = zs.gathered_dot(u, sut, vv, svt)
            else:
                gu = torch.index_select(u.tensor, 1, sut.long())
                gv = torch.index_select(vv.tensor, 1, svt.long())
                r_logits = (gu * gv).sum(2)
            bn.deterministic('r_pred', torch.sigmoid(r_logits))
            bn.normal('r', torch.sigmoid(r_logits), std=alpha_pred)
            return bn

        model = pmf()

        def log_joint(bn):
            log_pu, log_pv = bn.cond_log_prob(['u', 'v'])
            log_pr = bn.cond_log_prob('r')
            return log_pu.sum(-1) + log_pv.sum(-1) + log_pr.sum(-1)
        model.log_joint = log_joint
        return model

    kw = dict(step_size=2e-3, n_leapfrogs=10, adapt_step_size=None,
              target_acceptance_rate=0.9, seed=9)
    runs = {}
    for fused in (True, False):
        hmc = zs.HMC(**kw)
        q = T(u0.copy())
        op, info = hmc.sample(build(fused), {'r': rt, 'v': vt}, {'u': q})
        # both spellings are lowered to the native plan (round 4): no
        # autograd graph, the rating terms and their scatter in two launches
        assert hmc.plan_kind == 'gathered_dot', hmc.plan_reason
        runs[fused] = (hmc, q, op, info)
    ref = hmc_ref.HMC(**kw)
    qr = u0.copy()
    lj = lambda qs: pmf_ref.log_joint(qs[0], v, su, sv, r, alpha_u, alpha_v,
                                      alpha_pred)
    gr = lambda qs: [pmf_ref.grad_log_joint(qs[0], v, su, sv, r, alpha_u,
                                            alpha_v, alpha_pred)[0]]
    ref.sample(lj, gr, [qr])
    accs = []
    for it in range(6):
        rinfo = ref.step()
        for fused in (True, False):
            runs[fused][2].run()
        fi, di = runs[True][3], runs[False][3]
        np.testing.assert_allclose(fi.orig_log_prob.cpu().numpy(),
                                   di.orig_log_prob.cpu().numpy(), rtol=2e-5)
        np.testing.assert_allclose(fi.hamiltonian.cpu().numpy(),
                                   di.hamiltonian.cpu().numpy(), rtol=2e-5,
                                   atol=0.05)
        np.testing.assert_allclose(fi.orig_log_prob.cpu().numpy(),
                                   rinfo.orig_log_prob, rtol=1e-4)
        np.testing.assert_allclose(fi.acceptance_rate.cpu().numpy(),
                                   rinfo.acceptance_rate, atol=0.03)
        accs.append(float(fi.acceptance_rate.mean()))
        # keep the three samplers on the same state
        runs[False][1].copy_(runs[True][1])
        qr[...] = runs[True][1].cpu().numpy()
    assert np.mean(accs) > 0.3
    # the chain moves towards the data: likelihood term improves
    assert float(runs[True][3].log_prob.mean()) > float(
        pmf_ref.log_joint(u0, v, su, sv, r, alpha_u, alpha_v, alpha_pred).mean())


def test_empty_pair_list(env):
    zs, torch, dev = env
    u = torch.randn(3, 4, 5, device=dev, requires_grad=True)
    v = torch.randn(3, 6, 5, device=dev, requires_grad=True)
    e = torch.zeros(0, dtype=torch.int64, device=dev)
    out = zs.gathered_dot(u, e, v, e)
    assert tuple(out.shape) == (3, 0)
    out.sum().backward()
    assert not u.grad.any() and not v.grad.any()


def test_rating_likelihood_kernel_matches_float64(env):
    """zshmc_gather_dot_normal_lik: sum_e log N(r_e; sigmoid(<u, v>), alpha) +
    constant, and d/d logit, against float64 (pmf_hmc.py:26-31)."""
    zs, torch, dev = env
    from zhusuan_amd import _capi
    rng = np.random.RandomState(5)
    for K, n, m, D, E, obs_rows in ((6, 7, 5, 4, 20, 1), (3, 40, 60, 30, 1000, 3),
                                    (8, 10, 10, 17, 129, 1), (2, 3, 3, 5, 0, 1)):
        u = rng.normal(size=(K, n, D)).astype(np.float32) * 0.7
        v = rng.normal(size=(K, m, D)).astype(np.float32) * 0.7
        su = rng.randint(0, n, size=E).astype(np.int32)
        sv = rng.randint(0, m, size=E).astype(np.int32)
        r = rng.uniform(size=(obs_rows, E)).astype(np.float32)
        const = rng.normal(size=K).astype(np.float32)
        logstd = float(np.log(0.2))
        T = lambda a: torch.tensor(a, device=dev)
        ut, vt, sut, svt, rt, ct = T(u), T(v), T(su), T(sv), T(r), T(const)
        g = torch.zeros(K, max(E, 1), device=dev)
        ll = torch.empty(K, device=dev)
        ws = torch.empty(max(1, int(_capi.load().zshmc_gather_dot_normal_workspace(
            K, E))), device=dev)
        for _ in range(2):
            _capi.call('zshmc_gather_dot_normal_lik', ut.data_ptr(),
                       vt.data_ptr(), sut.data_ptr(), svt.data_ptr(),
                       rt.data_ptr(), obs_rows, logstd, ct.data_ptr(), K, n, m,
                       E, D, g.data_ptr(), ll.data_ptr(), ws.data_ptr(),
                       _capi.current_stream())
            first = (ll.clone(), g.clone()) if _ == 0 else first
        assert torch.equal(first[0], ll) and torch.equal(first[1], g)
        d = (u.astype(np.float64)[:, su] * v.astype(np.float64)[:, sv]).sum(-1)
        p = 1 / (1 + np.exp(-d))
        diff = (r.astype(np.float64) if obs_rows == K else
                r.astype(np.float64)[0][None]) - p
        prec = np.exp(-2 * logstd)
        ll_ref = (-0.5 * np.log(2 * np.pi) - logstd -
                  0.5 * prec * diff ** 2).sum(-1) + const
        np.testing.assert_allclose(ll.cpu().numpy(), ll_ref, rtol=2e-6,
                                   atol=1e-5 * max(1.0, np.abs(ll_ref).max()))
        if E:
            np.testing.assert_allclose(g.cpu().numpy()[:, :E].reshape(K, E),
                                       diff * prec * p * (1 - p), rtol=2e-5,
                                       atol=2e-5)


@pytest.mark.parametrize('side', ['u', 'v'])
@pytest.mark.parametrize('spelling', ['fused', 'gathers'])
@pytest.mark.parametrize('adapt', [False, True])
def test_native_gathered_dot_plan_follows_oracle_and_generic_plan(
        env, side, spelling, adapt):
    """The rating model on the NATIVE plan (csrc/gather_dot.hip +
    csrc/hmc_model_seg.hip: no autograd graph): sampling u given v and v given
    u, zs.gathered_dot and the reference's two gathers, with and without
    step-size + mass adaptation -- every transition against the oracle
    (oracle/pmf_ref.py under oracle/hmc_ref.py) and against the generic
    (autograd) plan on the same model."""
    zs, torch, dev = env
    rng = np.random.RandomState(11)
    K, n, m, D, E = 6, 14, 10, 6, 150
    su, sv, r, u0, v = _pmf_problem(rng, K, n, m, D, E)
    alpha_u, alpha_v, alpha_pred = 1.0, 0.8, 0.2
    T = lambda a, **kw: torch.tensor(a, device=dev, **kw)
    sut, svt, rt = T(su, dtype=torch.int32), T(sv, dtype=torch.int32), T(r)
    zeros_u, zeros_v = torch.zeros(n, D, device=dev), torch.zeros(m, D,
                                                                   device=dev)
    v0 = (v + 0.1 * rng.normal(size=v.shape)).astype(np.float32)
    fixed_u = (0.3 * rng.normal(size=(K, n, D))).astype(np.float32)

    def build():
        @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
        def pmf():
            bn = zs.BayesianNet()
            u = bn.normal('u', zeros_u, std=alpha_u, n_samples=K,
                          group_ndims=1)
            vv = bn.normal('v', zeros_v, std=alpha_v, n_samples=K,
                           group_ndims=1)
            if spelling == 'fused':
                r_logits = zs.gathered_dot(u, sut, vv, svt)
            else:                                   # pmf_hmc.py:26-28
                r_logits = (u.tensor[:, sut.long()] *
                            vv.tensor[:, svt.long()]).sum(2)
            bn.deterministic('r_pred', torch.sigmoid(r_logits))
            bn.normal('r', torch.sigmoid(r_logits), std=alpha_pred)
            return bn
        model = pmf()

        def log_joint(bn):
            log_pu, log_pv = bn.cond_log_prob(['u', 'v'])
            return log_pu.sum(-1) + log_pv.sum(-1) + \
                bn.cond_log_prob('r').sum(-1)
        model.log_joint = log_joint
        return model

    kw = dict(step_size=0.02, n_leapfrogs=6, seed=17)
    if adapt:
        kw.update(adapt_step_size=True, adapt_mass=True,
                  mass_collect_iters=2, target_acceptance_rate=0.8)
    lat, q0 = ('u', u0) if side == 'u' else ('v', v0)
    fixed_name, fixed = ('v', v) if side == 'u' else ('u', fixed_u)

    def sampler(native):
        q = T(q0.copy())
        hmc = zs.HMC(native_plans=native, **kw)
        op, info = hmc.sample(build(), {'r': rt, fixed_name: T(fixed)},
                              {lat: q})
        return hmc, op, info, q

    hmc, op, info, q = sampler(True)
    if side == 'v' and spelling == 'gathers':
        # the symbolic layer follows the gathers of the FIRST factor only
        # (`latent[:, idx] * const`); v as the latent is the second factor
        # here: `const * latent[:, idx]` -- also recognised (__rmul__)
        pass
    assert hmc.plan_kind == 'gathered_dot', hmc.plan_reason
    hg, opg, infog, qg = sampler(False)
    assert hg.plan_kind == 'generic'

    def lj(qs):
        uu, vv = (qs[0], v) if side == 'u' else (fixed_u, qs[0])
        return pmf_ref.log_joint(uu, vv, su, sv, r, alpha_u, alpha_v,
                                 alpha_pred)

    def gr(qs):
        uu, vv = (qs[0], v) if side == 'u' else (fixed_u, qs[0])
        return [pmf_ref.grad_log_joint(uu, vv, su, sv, r, alpha_u, alpha_v,
                                       alpha_pred)[0 if side == 'u' else 1]]
    ref = hmc_ref.HMC(**kw)
    qr = [q0.copy()]
    ref.sample(lj, gr, qr)
    for it in range(7):
        op.run()
        opg.run()
        rinfo = ref.step()
        for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob',
                  'log_prob'):
            want = getattr(rinfo, f)
            h = np.abs(want).max()
            np.testing.assert_allclose(getattr(info, f).cpu().numpy(), want,
                                       rtol=0, atol=3e-5 * h + 2e-3,
                                       err_msg='%s it %d' % (f, it))
            np.testing.assert_allclose(getattr(infog, f).cpu().numpy(),
                                       getattr(info, f).cpu().numpy(), rtol=0,
                                       atol=3e-5 * h + 2e-3)
        np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(),
                                   rinfo.acceptance_rate, atol=5e-3)
        np.testing.assert_allclose(float(info.updated_step_size.item()),
                                   float(ref.step_size), rtol=2e-3)
        agree = np.isclose(q.cpu().numpy(), qr[0], rtol=2e-3,
                           atol=2e-3).reshape(K, -1).all(1)
        assert agree.sum() >= K - 1
        q.copy_(T(qr[0]))
        qg.copy_(T(qr[0]))


def test_native_gathered_dot_plan_with_fed_minibatches(env):
    """pmf_hmc.py:84-87,186-192: the pair lists, the ratings and the observed
    factor table change from run to run (placeholders / deferred graph
    expressions); the native plan follows the feeds exactly as the generic
    plan does, and a table whose size is not a multiple of 4 is refused
    aloud."""
    zs, torch, dev = env
    rng = np.random.RandomState(3)
    K, n, m, D = 4, 10, 12, 6
    zeros_u, zeros_v = torch.zeros(n, D, device=dev), torch.zeros(m, D,
                                                                   device=dev)
    batches = []
    for E in (40, 75, 1):
        su, sv, r, _, v = _pmf_problem(rng, K, n, m, D, E)
        batches.append((su.astype(np.int32), sv.astype(np.int32), r, v))

    def run(native):
        dflt = lambda a: torch.tensor(a, device=dev)
        sel_u = zs.placeholder(torch.int32, name='su',
                               default=dflt(batches[0][0]))
        sel_v = zs.placeholder(torch.int32, name='sv',
                               default=dflt(batches[0][1]))
        rating = zs.placeholder(torch.float32, name='r',
                                default=dflt(batches[0][2] * 4 + 1))
        v_obs = zs.placeholder(torch.float32, name='v',
                               default=dflt(batches[0][3]))

        @zs.meta_bayesian_net(scope='pmf', reuse_variables=True)
        def pmf():
            bn = zs.BayesianNet()
            u = bn.normal('u', zeros_u, std=1.0, n_samples=K, group_ndims=1)
            vv = bn.normal('v', zeros_v, std=1.0, n_samples=K, group_ndims=1)
            lg = zs.gathered_dot(u, sel_u.value, vv, sel_v.value)
            bn.normal('r', torch.sigmoid(lg), std=0.25)
            return bn
        model = pmf()
        model.log_joint = lambda bn: (
            bn.cond_log_prob('u').sum(-1) + bn.cond_log_prob('v').sum(-1) +
            bn.cond_log_prob('r').sum(-1))
        q = torch.full((K, n, D), 0.05, device=dev)
        hmc = zs.HMC(step_size=0.03, n_leapfrogs=4, seed=2,
                     native_plans=native)
        observed = {'r': zs.deferred(lambda: (rating.value - 1.0) / 4.0),
                    'v': v_obs}
        op, info = hmc.sample(model, observed, {'u': q})
        out = []
        for i in range(9):
            su, sv, r, v = batches[i % 3]
            op.run(feed_dict={sel_u: su, sel_v: sv, rating: r * 4 + 1,
                              v_obs: v})
            out.append((info.log_prob.clone(), q.clone()))
        return hmc, out

    hn, a = run(True)
    hg, b = run(False)
    assert hn.plan_kind == 'gathered_dot', hn.plan_reason
    assert hg.plan_kind == 'generic'
    for (lpa, qa), (lpb, qb) in zip(a, b):
        torch.testing.assert_close(lpa, lpb, rtol=1e-5, atol=2e-3)
        same = torch.isclose(qa, qb, rtol=1e-3, atol=1e-3).reshape(
            K, -1).all(1)
        assert int(same.sum()) >= K - 1

    # 7 x 5 = 35 elements per chain: not a multiple of 4
    zu = torch.zeros(7, 5, device=dev)
    sel = torch.tensor([0, 1, 6], dtype=torch.int32, device=dev)

    @zs.meta_bayesian_net()
    def odd():
        bn = zs.BayesianNet()
        u = bn.normal('u', zu, std=1.0, n_samples=K, group_ndims=2)
        lg = zs.gathered_dot(u, sel, torch.ones(K, 3, 5, device=dev),
                             torch.tensor([0, 1, 2], dtype=torch.int32,
                                          device=dev))
        bn.normal('r', torch.sigmoid(lg), std=0.5, group_ndims=1)
        return bn
    hmc = zs.HMC(step_size=0.01)
    with pytest.warns(zs.NativePlanFallbackWarning, match='multiple of 4'):
        hmc.sample(odd(), {'r': torch.zeros(3, device=dev)},
                   {'u': torch.zeros(K, 7, 5, device=dev)})
    assert hmc.plan_kind == 'generic' and '35' in hmc.plan_reason


# -- likelihood + gradient in one pass (zshmc_gather_dot_normal_lik_grad) ------
@pytest.mark.parametrize('K,n_lat,n_other,E,D,per_chain_obs', [
    (8, 40, 30, 3000, 32, False),    # long rows: several 256-slot segments
    (5, 17, 9, 200, 4, True),        # chains not a multiple of 8, tiny rows
    (1, 300, 50, 1000, 36, False),   # two 32-float chunks, rows without pairs
    (3, 10, 10, 2600, 128, False),   # widest rows, every row long
    (8, 6, 5, 0, 32, False)])        # no pairs at all
def test_fused_likelihood_and_gradient_match_float64(K, n_lat, n_other, E, D,
                                                     per_chain_obs):
    """pmf_hmc.py:26-31 and what tf.gradients (hmc.py:430-432) gives for the
    latent table, from the one-pass kernel over the segmented CSR view."""
    import torch
    from zhusuan_amd import _capi, _ops
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(E + D)
    u = (0.4 * rng.normal(size=(K, n_lat, D))).astype(np.float32)
    v = (0.4 * rng.normal(size=(K, n_other, D))).astype(np.float32)
    # skewed: a few latent rows own most pairs; some own none
    su = np.minimum((rng.exponential(0.15, size=E) * n_lat).astype(np.int64),
                    n_lat - 1)
    sv = rng.randint(0, n_other, size=E)
    r = rng.uniform(size=(K if per_chain_obs else 1, E)).astype(np.float32)
    logstd = -0.7
    lp_const = rng.normal(size=K).astype(np.float32)
    ut, vt = torch.tensor(u, device=dev), torch.tensor(v, device=dev)
    sut = torch.tensor(su, device=dev)
    if E == 0:
        seg = torch.zeros(n_lat + 1, dtype=torch.int32, device=dev)
        order = torch.zeros(0, dtype=torch.long, device=dev)
    else:
        _, seg, order = _ops._pair_csr(sut, n_lat, 'test_fused')
        order = order.long()
    sp, sr, sf, lr = _ops._csr_segments(seg, E)
    assert int(sr.numel()) >= n_lat
    if E > 2000:
        assert lr.numel() > 0               # some rows really are cut
    idx_csr = torch.tensor(sv, device=dev, dtype=torch.int32)[order].contiguous()
    obs_csr = torch.tensor(r, device=dev)[:, order].contiguous()
    grad = torch.full((K, n_lat, D), float('nan'), device=dev)
    ll = torch.full((K,), float('nan'), device=dev)
    ws = torch.empty(K * int(sr.numel()) * (D + 1) + 4, device=dev)
    for rep in range(2):
        _capi.call('zshmc_gather_dot_normal_lik_grad', ut.data_ptr(),
                   vt.data_ptr(), sp.data_ptr(), sr.data_ptr(), sf.data_ptr(),
                   lr.data_ptr() if lr.numel() else None, lr.numel(),
                   idx_csr.data_ptr() if E else None,
                   obs_csr.data_ptr() if E else None, r.shape[0], logstd,
                   torch.tensor(lp_const, device=dev).data_ptr(), K, n_lat,
                   n_other, E, sr.numel(), D, grad.data_ptr(), ll.data_ptr(),
                   ws.data_ptr(), _capi.current_stream())
        torch.cuda.synchronize()
        if rep == 0:
            g0, l0 = grad.cpu().numpy().copy(), ll.cpu().numpy().copy()
    np.testing.assert_array_equal(grad.cpu().numpy(), g0)     # bit-stable
    np.testing.assert_array_equal(ll.cpu().numpy(), l0)
    u64, v64 = u.astype(np.float64), v.astype(np.float64)
    d = np.einsum('ked,ked->ke', u64[:, su], v64[:, sv]) if E else \
        np.zeros((K, 0))
    pred = 1 / (1 + np.exp(-d))
    prec = np.exp(-2 * logstd)
    diff = r.astype(np.float64) - pred
    ll_ref = (-0.5 * np.log(2 * np.pi) - logstd - 0.5 * prec * diff ** 2
              ).sum(1) + lp_const
    g = diff * prec * pred * (1 - pred)
    g_ref = np.zeros((K, n_lat, D))
    for k in range(K):
        np.add.at(g_ref[k], su, g[k][:, None] * v64[k, sv])
    np.testing.assert_allclose(l0, ll_ref, rtol=2e-5, atol=2e-4 * max(E, 1) ** .5)
    np.testing.assert_allclose(g0, g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))
