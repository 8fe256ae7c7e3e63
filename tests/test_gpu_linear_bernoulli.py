"""GPU parity of the fused dense-logit Bernoulli kernel
(csrc/linear_bernoulli.hip, fp32 MFMA) vs the oracle's materialised
Bernoulli(logits = w @ X^T) log_prob / gradient, and end-to-end HMC on a
Bayesian logistic regression (BASELINE config 3 at reduced size) through
`Bernoulli(linear_logits(w, X))`."""
import numpy as np
import pytest

from oracle import hmc_ref
from oracle.distributions_ref import Bernoulli as RB, Normal as RN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _data(C, N, D, seed):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(N, D)).astype(np.float32)
    w_true = rng.normal(size=D).astype(np.float32)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true / np.sqrt(D)))
         ).astype(np.int32)
    W = (rng.normal(size=(C, D)) * 0.3).astype(np.float32)
    return X, y, W


def _ref(W, X, y):
    l = W.astype(np.float64) @ X.astype(np.float64).T
    yf = y.astype(np.float64)
    ll = (yf * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum(1)
    g = (yf - 1 / (1 + np.exp(-l))) @ X.astype(np.float64)
    return ll, g


# ragged C (not a multiple of 64), ragged N (not a multiple of 32), every
# kernel width (64/128/256) and zero-padded feature counts
@pytest.mark.parametrize('C,N,D', [(64, 32, 64), (100, 1000, 256), (7, 45, 128),
                                   (130, 333, 20), (64, 4096, 200), (1, 1, 3),
                                   (256, 10000, 256),
                                   # the 192-wide instantiation (b64 operand
                                   # reads, 12-byte DMA lanes)
                                   (100, 777, 150), (64, 64, 192), (3, 130, 129),
                                   # the 16-chain-block kernel (320 .. 576:
                                   # csrc/linear_bernoulli_mid.hip): every
                                   # width, ragged 64-chain blocks, ragged
                                   # 16-row tiles
                                   (100, 1000, 512), (33, 77, 300),
                                   (70, 500, 384), (20, 130, 450), (65, 200, 576),
                                   (5, 17, 520), (130, 16, 321), (1, 15, 448),
                                   (40, 100, 600), (66, 70, 768), (10, 33, 890),
                                   # the feature-split kernel (1024): ragged
                                   # 32-chain blocks, ragged 32-row tiles,
                                   # one-row and one-chain shapes
                                   (70, 2051, 1024), (64, 333, 700), (1, 1, 257),
                                   (31, 32, 1000)])
def test_loglik_and_grad_match_float64_reference(env, C, N, D):
    zs, torch, dev = env
    X, y, W = _data(C, N, D, seed=C + N + D)
    wt = torch.tensor(W, device=dev, requires_grad=True)
    d = zs.distributions.Bernoulli(
        zs.linear_logits(wt, torch.tensor(X, device=dev)), group_ndims=1)
    ll = d.log_prob(torch.tensor(y, device=dev))
    assert tuple(ll.shape) == (C,)
    ll_ref, g_ref = _ref(W, X, y)
    # fp32 MFMA = k-ordered fmaf chain: error ~1e-7 * sum|terms|
    np.testing.assert_allclose(ll.detach().cpu().numpy(), ll_ref,
                               rtol=2e-5, atol=2e-5 * N)
    coef = torch.linspace(0.5, 1.5, C, device=dev)
    (ll * coef).sum().backward()
    scale = np.abs(g_ref).max() + 1.0
    np.testing.assert_allclose(wt.grad.cpu().numpy(),
                               g_ref * coef.cpu().numpy()[:, None],
                               rtol=1e-4, atol=2e-5 * scale)
    # and agrees with the element-wise HIP Bernoulli kernel on dense logits
    dense = zs.distributions.Bernoulli(wt.detach() @ torch.tensor(
        X, device=dev).t(), group_ndims=1).log_prob(torch.tensor(y, device=dev))
    np.testing.assert_allclose(ll.detach().cpu().numpy(), dense.cpu().numpy(),
                               rtol=2e-5, atol=2e-5 * N)


def test_multi_axis_chains_and_fallback(env):
    zs, torch, dev = env
    X, y, W = _data(24, 50, 10, seed=1)
    w3 = torch.tensor(W.reshape(4, 6, 10), device=dev)
    Xt, yt = torch.tensor(X, device=dev), torch.tensor(y, device=dev)
    ll = zs.distributions.Bernoulli(zs.linear_logits(w3, Xt),
                                    group_ndims=1).log_prob(yt)
    assert tuple(ll.shape) == (4, 6)
    ll_ref, _ = _ref(W, X, y)
    np.testing.assert_allclose(ll.cpu().numpy().reshape(-1), ll_ref, rtol=2e-5,
                               atol=1e-3)
    # group_ndims = 0 cannot use the fused kernel: dense fallback, [4, 6, 50]
    e = zs.distributions.Bernoulli(zs.linear_logits(w3, Xt)).log_prob(yt)
    assert tuple(e.shape) == (4, 6, 50)
    np.testing.assert_allclose(e.sum(-1).cpu().numpy().reshape(-1), ll_ref,
                               rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize('native,D', [(True, 16), (False, 16), (True, 64),
                                      (True, 20)])
def test_bayesian_logistic_regression_hmc(env, native, D):
    """w ~ N(0, 1), y ~ Bernoulli(w X^T) vs the oracle with materialised
    logits: the native plan (fused MFMA likelihood + csrc/hmc_model.hip, no
    autograd; D = 64: no operand padding, D = 20: padded) and the generic
    plan (autograd glue around the same likelihood kernel)."""
    zs, torch, dev = env
    C, N = 96, 600
    X, y, W0 = _data(C, N, D, seed=5)
    W0 *= 0.3
    Xt, yt = torch.tensor(X, device=dev), torch.tensor(y, device=dev)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D, device=dev),
                      std=torch.ones(D, device=dev), n_samples=C,
                      group_ndims=1)
        bn.bernoulli('y', zs.linear_logits(w.tensor, Xt), group_ndims=1)
        return bn

    wt = torch.tensor(W0, device=dev)
    hmc = zs.HMC(step_size=0.01, n_leapfrogs=6, adapt_step_size=True, seed=9,
                 native_plans=native)
    op, info = hmc.sample(blr(), {'y': yt}, {'w': wt})
    assert hmc.plan_kind == ('linear_bernoulli' if native else 'generic')

    def lj(q):
        w = q[0]
        return (RN(np.zeros(D, np.float32), std=np.ones(D, np.float32),
                   group_ndims=1).log_prob(w) +
                RB((w @ X.T).astype(np.float32), group_ndims=1).log_prob(y))

    def grad(q):
        w = q[0]
        l = (w @ X.T).astype(np.float32)
        res = y.astype(np.float32) - 1 / (1 + np.exp(-l))
        return [(-w + res @ X).astype(np.float32)]

    wr = W0.copy()
    ref = hmc_ref.HMC(step_size=0.01, n_leapfrogs=6, adapt_step_size=True,
                      seed=9)
    ref.sample(lj, grad, [wr])
    for it in range(4):
        rinfo = ref.step()
        op.run()
        np.testing.assert_allclose(info.orig_log_prob.cpu().numpy(),
                                   rinfo.orig_log_prob, rtol=3e-5, atol=5e-3)
        np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(),
                                   rinfo.acceptance_rate, atol=1e-2)
        np.testing.assert_allclose(float(info.updated_step_size.item()),
                                   float(rinfo.updated_step_size), rtol=1e-2)
        ok = np.abs(ref.last_u01 - rinfo.acceptance_rate) > 2e-2
        np.testing.assert_allclose(wt.cpu().numpy()[ok], wr[ok], atol=5e-4)
        wt.copy_(torch.tensor(wr, device=dev))


def test_config3_full_size_properties(env):
    """BASELINE config 3 at its full size (32 768 chains, 10^6 rows, D = 256;
    logits would be 131 GB): additivity of the likelihood and its gradient
    over a split of the rows, and a few chains against float64 NumPy."""
    zs, torch, dev = env
    from zhusuan_amd import _capi
    C, N, D = 32768, 1000000, 256
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N, D, device=dev, generator=g)
    w_true = torch.randn(D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) <
         torch.sigmoid(X @ w_true / D ** 0.5)).float()
    W = torch.randn(C, D, device=dev, generator=g) * 0.05
    s = torch.cuda.current_stream().cuda_stream

    def run(x, yy):
        ll = torch.empty(C, device=dev)
        gw = torch.empty(C, D, device=dev)
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), x.data_ptr(),
                   yy.data_ptr(), C, x.shape[0], D, ll.data_ptr(),
                   gw.data_ptr(), 1, None, s)
        return ll, gw

    ll, gw = run(X, y)
    h = 400000 + 37                       # ragged split (not a tile multiple)
    ll_a, gw_a = run(X[:h].contiguous(), y[:h].contiguous())
    ll_b, gw_b = run(X[h:].contiguous(), y[h:].contiguous())
    assert bool(torch.isfinite(ll).all()) and bool(torch.isfinite(gw).all())
    torch.testing.assert_close(ll_a + ll_b, ll, rtol=2e-5, atol=0.5)
    scale = float(gw.abs().max())
    torch.testing.assert_close(gw_a + gw_b, gw, rtol=1e-4, atol=2e-5 * scale)
    # three chains against float64
    idx = [0, 12345, C - 1]
    Xh, yh = X.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64)
    for c in idx:
        w = W[c].cpu().numpy().astype(np.float64)
        l = Xh @ w
        ll_ref = (yh * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum()
        g_ref = (yh - 1 / (1 + np.exp(-l))) @ Xh
        np.testing.assert_allclose(float(ll[c]), ll_ref, rtol=2e-5)
        np.testing.assert_allclose(gw[c].cpu().numpy(), g_ref, rtol=1e-4,
                                   atol=2e-5 * np.abs(g_ref).max())


@pytest.mark.parametrize('D', [64, 192, 256, 320, 512, 832, 1024])
def test_row_range_splits_match_single_pass(D):
    """n_splits > 1 (small chain counts) is the same sum in a different, fixed
    association: equal to the unsplit launch within fp32 re-association, and
    bit-identical run to run."""
    import torch
    from zhusuan_amd import _capi
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(11)
    C, N = 130, 5037
    X = torch.randn(N, D, device=dev, generator=g)
    y = (torch.rand(N, device=dev, generator=g) < 0.4).float()
    W = torch.randn(C, D, device=dev, generator=g) * 0.1
    s = torch.cuda.current_stream().cuda_stream

    def run(splits, grad=True):
        ll = torch.empty(C, device=dev)
        gw = torch.empty(C, D, device=dev) if grad else None
        ws = torch.empty(splits * C * (D + 1), device=dev) if splits > 1 else None
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
                   y.data_ptr(), C, N, D, ll.data_ptr(), _capi.ptr(gw), splits,
                   _capi.ptr(ws), s)
        return ll, gw

    ll1, g1 = run(1)
    z = (W.double() @ X.double().t())
    want = (y.double() * z - torch.nn.functional.softplus(z)).sum(-1)
    torch.testing.assert_close(ll1.double(), want, rtol=1e-5, atol=1e-2)
    for splits in (2, 3, 7, 16):
        ll, gw = run(splits)
        torch.testing.assert_close(ll, ll1, rtol=1e-5, atol=1e-2)
        torch.testing.assert_close(gw, g1, rtol=1e-4, atol=1e-3)
        ll_b, gw_b = run(splits)
        assert torch.equal(ll, ll_b) and torch.equal(gw, gw_b)
        ll_n, _ = run(splits, grad=False)
        torch.testing.assert_close(ll_n, ll1, rtol=1e-5, atol=1e-2)
        # gradient only (log_lik = NULL), the form a trajectory's interior
        # evaluations take: its own instantiation (no row masks, zeroed tail)
        gw_o = torch.empty(C, D, device=dev)
        ws = torch.empty(splits * C * (D + 1), device=dev)
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
                   y.data_ptr(), C, N, D, None, gw_o.data_ptr(), splits,
                   ws.data_ptr(), s)
        torch.testing.assert_close(gw_o, g1, rtol=1e-4, atol=1e-3)
    # more splits than tiles: the trailing row ranges are empty and contribute
    # zeros (a caller of the C-ABI may ask for this; the front-end never does)
    n_few = 70
    ll_f = torch.empty(C, device=dev)
    gw_f = torch.empty(C, D, device=dev)
    ws = torch.empty(64 * C * (D + 1), device=dev)
    _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
               y.data_ptr(), C, n_few, D, ll_f.data_ptr(), gw_f.data_ptr(), 64,
               ws.data_ptr(), s)
    z = (W.double() @ X[:n_few].double().t())
    torch.testing.assert_close(
        ll_f.double(), (y[:n_few].double() * z -
                        torch.nn.functional.softplus(z)).sum(-1),
        rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(
        gw_f.double(), (y[:n_few].double() - torch.sigmoid(z)) @
        X[:n_few].double(), rtol=1e-4, atol=1e-3)
    with pytest.raises(_capi.ZshmcError):
        ll = torch.empty(C, device=dev)
        _capi.call('zshmc_linear_bernoulli_log_lik', W.data_ptr(), X.data_ptr(),
                   y.data_ptr(), C, N, D, ll.data_ptr(), None, 4, None, s)
