"""csrc/sparse_multinomial.hip (ABI 0.6.0): the topic model's likelihood and
gradient ROW BY ROW over each row's own words, exact float32 on the vector
ALU -- the small-problem form (lntm_mcem.py:62-70,157-182 runs one chain x a
minibatch of 100 documents).  Against float64 (multivariate.py:435-443 on the
materialised log(theta @ phi); hmc.py:430-432), against the fp32 MFMA kernel
on the same operands, and through HMC: the plan takes it for small sparse
problems on the fp32 path, sample_op.run and run_many agree bit for bit, and
the trajectory follows the MFMA kernels' within float32 summation noise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from zhusuan_amd import _capi
    assert torch.cuda.is_available()
    return torch, _capi, torch.device('cuda', 0)


def _data(n_chains, n_docs, V, K, rate, seed):
    rng = np.random.RandomState(seed)
    eta = rng.normal(size=(n_chains * n_docs, K))
    theta = np.exp(eta - eta.max(1, keepdims=True))
    theta = (theta / theta.sum(1, keepdims=True)).astype(np.float32)
    beta = rng.normal(size=(K, V))
    phi = np.exp(beta - beta.max(1, keepdims=True))
    phi = (phi / phi.sum(1, keepdims=True)).astype(np.float32)
    x = rng.poisson(rate, size=(n_docs, V)).astype(np.float32)
    return theta, phi, x


def _ref(theta, phi, x, n_docs):
    S = theta.astype(np.float64) @ phi.astype(np.float64)
    R = theta.shape[0]
    xr = x[np.arange(R) % n_docs].astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        ll = np.where(xr != 0, xr * np.log(S), 0.0).sum(1)
        g = np.where(xr != 0, xr / S, 0.0) @ phi.astype(np.float64).T
    return ll, g


@pytest.mark.parametrize('n_chains,n_docs,V,K,width,n_splits,rate', [
    (1, 100, 12419, 100, 128, 4, 0.08), (1, 100, 12419, 100, 128, 1, 0.08),
    (3, 7, 333, 20, 64, 2, 0.2), (2, 5, 500, 150, 192, 3, 0.1),
    (1, 9, 1000, 256, 256, 1, 0.05), (5, 1, 77, 64, 64, 1, 0.5),
    (4, 3, 2000, 128, 128, 7, 0.02)])
@pytest.mark.parametrize('want_ll', [True, False])
def test_rows_match_float64_and_the_mfma_kernel(env, n_chains, n_docs, V, K,
                                                width, n_splits, rate,
                                                want_ll):
    torch, _capi, dev = env
    from zhusuan_amd import _ops
    theta, phi, x = _data(n_chains, n_docs, V, K, rate, seed=V + K)
    if n_docs > 2:
        x[1] = 0.0                                  # a document without words
    R = n_chains * n_docs
    th = torch.zeros(R, width, device=dev)
    th[:, :K] = torch.tensor(theta, device=dev)
    phi_t = _ops._padded_phi_t(torch.tensor(phi, device=dev), width)
    xt = torch.tensor(x, device=dev)
    vals, rows, off, total = _ops.counts_csr(xt)
    ws = torch.empty(n_splits * R * (width + 1), device=dev) \
        if n_splits > 1 else None
    out = []
    for rep in range(2):
        ll = torch.full((R,), float('nan'), device=dev) if want_ll else None
        g = torch.full((R, width), float('nan'), device=dev)
        _capi.call('zshmc_sparse_multinomial_log_lik', th.data_ptr(),
                   phi_t.data_ptr(), vals.data_ptr(), rows.data_ptr(),
                   off.data_ptr(), n_docs, R, V, width, _capi.ptr(ll),
                   g.data_ptr(), n_splits, _capi.ptr(ws),
                   _capi.current_stream())
        torch.cuda.synchronize()
        out.append((None if ll is None else ll.cpu().numpy(),
                    g.cpu().numpy()))
    np.testing.assert_array_equal(out[0][1], out[1][1])        # bit-stable
    ll_ref, g_ref = _ref(theta, phi, x, n_docs)
    if want_ll:
        np.testing.assert_array_equal(out[0][0], out[1][0])
        np.testing.assert_allclose(out[0][0], ll_ref, rtol=2e-5,
                                   atol=2e-5 * V)
    np.testing.assert_allclose(out[0][1][:, :K], g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))
    assert (out[0][1][:, K:] == 0).all()            # the padding columns
    # the fp32 MFMA kernel on the same operands
    xp, stride = _ops._padded_counts(xt, 32)
    g2 = torch.full((R, width), float('nan'), device=dev)
    _capi.call('zshmc_linear_multinomial_log_lik', th.data_ptr(),
               phi_t.data_ptr(), xp.data_ptr(), n_docs, stride, R, V, width,
               None, g2.data_ptr(), 1, None, _capi.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0][1], g2.cpu().numpy(), rtol=2e-5,
                               atol=2e-6 * (np.abs(g_ref).max() + 1))


def test_refuses_what_it_does_not_take(env):
    torch, _capi, dev = env
    t = torch.zeros(64, 128, device=dev)
    i32 = torch.zeros(64, dtype=torch.int32, device=dev)
    off = torch.tensor([0, 32, 64], device=dev)

    def call(**kw):
        a = dict(theta=t, n_docs=2, R=4, K=128, grad=t, n_splits=1, ws=None)
        a.update(kw)
        _capi.call('zshmc_sparse_multinomial_log_lik', a['theta'].data_ptr(),
                   t.data_ptr(), t.data_ptr(), i32.data_ptr(), off.data_ptr(),
                   a['n_docs'], a['R'], 100, a['K'], None,
                   _capi.ptr(a['grad']), a['n_splits'], a['ws'],
                   _capi.current_stream())
    call()
    with pytest.raises(_capi.ZshmcError, match='bad shape'):
        call(K=96)
    with pytest.raises(_capi.ZshmcError, match='bad shape'):
        call(R=5)                      # rows not whole copies of the documents
    with pytest.raises(_capi.ZshmcError, match='null pointer'):
        call(grad=None)
    with pytest.raises(_capi.ZshmcError, match='workspace'):
        call(n_splits=3)


def test_hmc_small_topic_model_runs_row_by_row(env, monkeypatch):
    """The E-step family at a small size on the default arithmetic: the plan
    picks the row-by-row kernel (sparse counts, few rows), run = run_many bit
    for bit (zshmc_hmc_model_run's dispatch through obs_sp_*), and the states
    follow the MFMA kernel's (SPARSE_ROWS_MAX = 0) within summation noise."""
    torch, _capi, dev = env
    import zhusuan_amd as zs
    rng = np.random.RandomState(5)
    n_chains, n_docs, K, V = 2, 24, 20, 1500
    phi = torch.softmax(torch.tensor(rng.normal(size=(K, V)).astype(
        np.float32), device=dev), -1)
    x = torch.tensor(rng.poisson(0.05, size=(n_docs, V)).astype(np.float32),
                     device=dev)
    eta0 = (0.3 * rng.normal(size=(n_chains, n_docs, K))).astype(np.float32)
    mean = torch.zeros(n_docs, K, device=dev)
    out = {}
    for mode, rows_max in (('loop', 4096), ('block', 4096), ('loop', 0)):
        monkeypatch.setattr(zs._ops, 'SPARSE_ROWS_MAX', rows_max)
        zs._ops.clear_caches()

        @zs.meta_bayesian_net(scope='lntm')
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', mean, logstd=0., n_samples=n_chains,
                            group_ndims=1)
            bn.unnormalized_multinomial(
                'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
                normalize_logits=False, dtype=torch.float32)
            return bn
        m = lntm()
        m.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
        flag = zs.placeholder(bool)
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=5, seed=9,
                     adapt_step_size=flag)
        eta = torch.tensor(eta0, device=dev)
        op, info = hmc.sample(m, {'x': x}, {'eta': eta})
        assert hmc.plan_kind == 'mixture_multinomial'
        assert hmc.likelihood_arithmetic_used == 'fp32'
        assert hmc._plan.sparse_rows == (rows_max > 0)
        if rows_max > 0:
            assert hmc._plan.n_inner_run < V // 4
        for n, feed in ((4, {flag: True}), (3, {flag: False})):
            if mode == 'block':
                op.run_many(n, feed_dict=feed)
            else:
                for _ in range(n):
                    op.run(feed_dict=feed)
        out[(mode, rows_max)] = (eta.cpu().numpy(), info.log_prob.cpu().numpy(),
                                 float(info.updated_step_size.item()))
    a, b, c = out[('loop', 4096)], out[('block', 4096)], out[('loop', 0)]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2] == b[2]
    close = np.isclose(a[0], c[0], atol=2e-4).reshape(n_chains * n_docs,
                                                      -1).all(1)
    assert close.mean() > 0.95, close.mean()       # a borderline accept flips
    np.testing.assert_allclose(a[2], c[2], rtol=1e-3)


def test_auto_keeps_a_one_chain_corpus_on_the_row_by_row_kernel(env):
    """lntm_mcem.py's layout at a size the matrix cores would take (5e10 flop
    per evaluation): one chain x 8 192 documents.  The bf16x3 alternative is
    the packed-rows form, which the row-by-row fp32 kernel beats up to ~16 000
    rows (0.335 against 0.77 ms per launch here): 'auto' stays on exact fp32
    and says why; 'bf16x3' asked for by name is honoured."""
    torch, _capi, dev = env
    import zhusuan_amd as zs
    g = torch.Generator(device=dev).manual_seed(0)
    n_docs, K, V = 8192, 128, 12419
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 0.08, device=dev), generator=g)
    mean = torch.zeros(n_docs, K, device=dev)

    def sampler(**kw):
        @zs.meta_bayesian_net(scope='lntm')
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', mean, logstd=0., n_samples=1, group_ndims=1)
            bn.unnormalized_multinomial(
                'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
                normalize_logits=False, dtype=torch.float32)
            return bn
        m = lntm()
        m.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
        hmc = zs.HMC(step_size=0.01, n_leapfrogs=2, seed=1, **kw)
        eta = torch.zeros(1, n_docs, K, device=dev)
        op, info = hmc.sample(m, {'x': x}, {'eta': eta})
        op.run()
        assert bool(torch.isfinite(info.log_prob).all())
        return hmc, info.log_prob.cpu().numpy()

    auto, lp_a = sampler()
    assert auto.likelihood_arithmetic_used == 'fp32'
    assert auto._plan.sparse_rows and 'row-by-row' in auto.arithmetic_reason
    b3, lp_b = sampler(likelihood_arithmetic='bf16x3')
    assert b3.likelihood_arithmetic_used == 'bf16x3'
    assert b3._plan.packed_rows and not b3._plan.sparse_rows
    same = np.isclose(lp_a, lp_b, rtol=2e-5, atol=2e-2)
    assert same.mean() > 0.99            # (a borderline accept may flip)
