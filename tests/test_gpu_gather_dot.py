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
