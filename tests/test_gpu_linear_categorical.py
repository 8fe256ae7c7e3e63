"""Dense-logit Categorical (softmax regression) on the fused fp32-MFMA kernels
(OP = 2 of csrc/linear_bernoulli.hip / linear_bernoulli_wide.hip, the class
softmax over accumulator lanes: csrc/lb_ops.h), the segmented kick/drift
kernel (csrc/hmc_model_seg.hip) and the 'linear_categorical' native plan.

  * kernel: log-likelihood and gradient against a float64 reference over
    class counts 2..32 (class strides 2, 4, 8, 16, 32, with and without
    padding classes), every kernel width (64..1024), ragged row tiles and
    chain blocks, row-range splits bit-stable;
  * zshmc_model_kick_drift_seg against a NumPy restatement (hmc.py:38-43 +
    univariate.py:174-181), feature counts that are and are not multiples of
    4, long rows (the PMF shape: few chains, 10^4 elements each);
  * plan: a free run equals the generic (autograd) plan's, and the oracle's
    transition by transition with step-size and mass adaptation on.
Reference: zhusuan/distributions/univariate.py:496-548 (Categorical._log_prob)
through tf.matmul(X, w, transpose_b=True), hmc.py:430-432 (its gradient)."""
import numpy as np
import pytest

from oracle import hmc_ref
from oracle.distributions_ref import Categorical as RC, Normal as RN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _ref64(w, X, y):
    """ll [C] and d ll / d w [C, K, F] in float64."""
    w, X = w.astype(np.float64), X.astype(np.float64)
    logits = np.einsum('nf,ckf->cnk', X, w)
    m = logits.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(-1))
    ll = (np.take_along_axis(logits, y[None, :, None].astype(np.int64),
                             -1)[..., 0] - lse).sum(-1)
    p = np.exp(logits - lse[..., None])
    res = -p
    res[:, np.arange(len(y)), y] += 1.0
    return ll, np.einsum('cnk,nf->ckf', res, X)


SHAPES = [
    # C, K, F, N
    (10, 4, 5, 30),        # the reference trace's shape
    (70, 2, 64, 130),      # stride 2, no padding anywhere
    (33, 3, 17, 257),      # padding class + padding columns
    (40, 10, 100, 515),    # MNIST-like class count
    (24, 16, 128, 300),
    (20, 8, 150, 333),     # the 192-wide instantiation
    (9, 32, 256, 96),      # a full half-wave of classes
    (12, 20, 300, 200),    # wide kernel (512), 12 padding classes
    (12, 10, 300, 200),    # 16-chain-block kernel (320), stride 16
    (6, 16, 500, 150),     # 16-chain-block kernel (512), a full row of classes
    (9, 3, 400, 77),       # (448), stride 4, one padding class
    (6, 10, 784, 150),     # wide kernel (1024)
    (130, 5, 8, 64),
]


@pytest.mark.parametrize('C,K,F,N', SHAPES)
def test_kernel_matches_float64_reference(env, C, K, F, N):
    zs, torch, dev = env
    from zhusuan_amd import _ops
    rng = np.random.RandomState(C * 1000 + K)
    X = rng.normal(size=(N, F)).astype(np.float32)
    w = (rng.normal(size=(C, K, F)) / np.sqrt(F)).astype(np.float32)
    w[0] *= 30.0                       # one chain with |logits| ~ 30
    y = rng.randint(0, K, size=N).astype(np.int32)
    ll_ref, g_ref = _ref64(w, X, y)
    wt = torch.tensor(w, device=dev, requires_grad=True)
    labels = _ops.labels_as_float(torch.tensor(y, device=dev), K)
    ll = _ops.LinearCategoricalLogLik.apply(wt, torch.tensor(X, device=dev),
                                            labels)
    ll.sum().backward()
    scale = np.abs(ll_ref).max()
    np.testing.assert_allclose(ll.detach().cpu().numpy(), ll_ref,
                               rtol=2e-6, atol=2e-6 * scale + 1e-4)
    gs = np.abs(g_ref).max()
    np.testing.assert_allclose(wt.grad.cpu().numpy(), g_ref, rtol=0,
                               atol=2e-5 * gs + 1e-5)
    # likelihood only (no gradient requested): the same values
    with torch.no_grad():
        ll2 = _ops.LinearCategoricalLogLik.apply(
            wt.detach(), torch.tensor(X, device=dev), labels)
    assert torch.equal(ll2, ll.detach())


def test_row_range_splits_are_deterministic_and_close(env):
    zs, torch, dev = env
    from zhusuan_amd import _capi, _ops
    C, K, F, N = 8, 10, 100, 6000
    G, width = _ops.class_stride(K), 128
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.zeros(N, width, device=dev)
    X[:, :F] = torch.randn(N, F, device=dev, generator=g)
    w = torch.zeros(C, G, width, device=dev)
    w[:, :K, :F] = torch.randn(C, K, F, device=dev, generator=g) / F ** 0.5
    y = torch.randint(0, K, (N,), device=dev, generator=g).float()
    stream = _capi.current_stream()

    def run(splits):
        ll = torch.empty(C * G, device=dev)
        gw = torch.empty(C * G, width, device=dev)
        ws = torch.empty(splits * C * G * (width + 1), device=dev) \
            if splits > 1 else None
        _capi.call('zshmc_linear_categorical_log_lik', w.data_ptr(),
                   X.data_ptr(), y.data_ptr(), C * G, N, width, K, G,
                   ll.data_ptr(), gw.data_ptr(), splits, _capi.ptr(ws),
                   stream)
        return ll, gw
    a, b = run(4), run(4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    one = run(1)
    torch.testing.assert_close(a[0].view(C, G).sum(-1),
                               one[0].view(C, G).sum(-1), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(a[1], one[1], rtol=1e-4, atol=1e-3)
    # padding classes: zero gradient rows, no log-likelihood terms
    assert float(a[1].view(C, G, width)[:, K:].abs().max()) == 0.0
    assert float(a[0].view(C, G)[:, K:].abs().max()) == 0.0


@pytest.mark.parametrize('C,K,F,G,stride', [
    (50, 4, 5, 4, 64),        # F % 4 != 0: element-wise addressing
    (20, 3, 64, 4, 64),       # a padding class row
    (7, 1, 20000, 1, 20000),  # the PMF shape: few chains, long rows, G = 1
    (300, 10, 100, 16, 128),
])
@pytest.mark.parametrize('with_mass', [False, True])
def test_segmented_kick_drift_matches_numpy(env, C, K, F, G, stride,
                                            with_mass):
    zs, torch, dev = env
    from zhusuan_amd import _capi
    rng = np.random.RandomState(K * 7 + F)
    D = K * F
    ld = (D + 3) // 4 * 4
    q = np.zeros((C, ld), np.float32)
    p = np.zeros((C, ld), np.float32)
    q[:, :D] = rng.normal(size=(C, D))
    p[:, :D] = rng.normal(size=(C, D))
    mean = np.zeros((2, ld), np.float32)
    logstd = np.zeros((1, ld), np.float32)
    mean[:, :D] = rng.normal(size=(2, D))
    logstd[:, :D] = 0.3 * rng.normal(size=(1, D))
    mass = np.ones(ld, np.float32)
    if with_mass:
        mass[:D] = np.exp(0.5 * rng.normal(size=D))
    grad = rng.normal(size=(C * G, stride)).astype(np.float32)
    ll = rng.normal(size=C * G).astype(np.float32)
    kin0 = rng.normal(size=C).astype(np.float32)
    eps, kick, drift, lik_scale = 0.07, 0.5, 1.0, 0.8
    t = lambda a: torch.tensor(a, device=dev)
    qd, pd, gd = t(q), t(p), t(grad)
    operand = torch.zeros(C * G, stride, device=dev)
    lp = torch.empty(C, device=dev)
    kin = t(kin0)
    ws = torch.empty(int(_capi.load().zshmc_model_seg_workspace(C, D)),
                     device=dev)
    md = t(mass)
    args = lambda: (
        qd.data_ptr(), pd.data_ptr(), gd.data_ptr(), stride, F, G,
        operand.data_ptr(), stride, t_mean.data_ptr(), 2, t_ls.data_ptr(), 1,
        md.data_ptr() if with_mass else None, None, eps, kick, drift,
        lik_scale, C, D, ld, t_ll.data_ptr(), lp.data_ptr(), kin.data_ptr(),
        ws.data_ptr(), _capi.current_stream())
    t_mean, t_ls, t_ll = t(mean), t(logstd), t(ll)
    _capi.call('zshmc_model_kick_drift_seg', *args())
    # NumPy restatement
    g_lik = grad.reshape(C, G, stride)[:, :K, :F].reshape(C, D)
    mu = mean[np.arange(C) % 2][:, :D]
    ls = logstd[0, :D]
    prec = np.exp(-2 * ls)
    r = q[:, :D] - mu
    prior = (-0.9189385332046727 - ls - 0.5 * prec * r * r).sum(1)
    g = lik_scale * g_lik - prec * r
    p1 = p[:, :D] + kick * eps * g
    q1 = q[:, :D] + drift * eps * p1 / mass[:D]
    np.testing.assert_allclose(pd.cpu().numpy()[:, :D], p1, rtol=2e-6,
                               atol=2e-6)
    np.testing.assert_allclose(qd.cpu().numpy()[:, :D], q1, rtol=2e-6,
                               atol=2e-6)
    lp_ref = lik_scale * ll.reshape(C, G).sum(1) + prior
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=3e-6,
                               atol=3e-6 * np.abs(lp_ref).max())
    kin_ref = kin0 + 0.5 * (p1 * p1 / mass[:D]).sum(1)
    np.testing.assert_allclose(kin.cpu().numpy(), kin_ref, rtol=3e-6,
                               atol=3e-6 * np.abs(kin_ref).max())
    op = operand.cpu().numpy().reshape(C, G, stride)
    np.testing.assert_array_equal(op[:, :K, :F].reshape(C, D),
                                  qd.cpu().numpy()[:, :D])
    assert np.abs(op[:, K:]).max(initial=0.0) == 0.0
    assert np.abs(op[:, :, F:]).max(initial=0.0) == 0.0
    # run-to-run bit stability (no atomics)
    lp_a, kin_a = lp.clone(), kin.clone()
    qd.copy_(t(q)), pd.copy_(t(p)), kin.copy_(t(kin0))
    _capi.call('zshmc_model_kick_drift_seg', *args())
    assert torch.equal(lp, lp_a) and torch.equal(kin, kin_a)


def _softmax_problem(C, K, F, N, seed):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(N, F)).astype(np.float32)
    w_true = rng.normal(size=(K, F)).astype(np.float32)
    y = np.argmax(X @ w_true.T + rng.gumbel(size=(N, K)), -1).astype(np.int32)
    w0 = (0.1 * rng.normal(size=(C, K, F))).astype(np.float32)
    return X, y, w0


def _oracle_model(X, y, std):
    def parts(w):
        prior = RN(np.float32(0), std=np.float32(std), group_ndims=2)
        logits = np.einsum('nf,ckf->cnk', X, w).astype(np.float32)
        return prior, RC(logits, group_ndims=1)

    def lj(qs):
        prior, lik = parts(qs[0])
        return (prior.log_prob(qs[0]) + lik.log_prob(y)).astype(np.float32)

    def grad(qs):
        prior, lik = parts(qs[0])
        return [(prior.grad_given(qs[0]) + np.einsum(
            'cnk,nf->ckf', lik.grad_logits(y), X)).astype(np.float32)]
    return lj, grad


@pytest.mark.parametrize('arith', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('C,K,F,N', [(48, 4, 5, 60), (40, 10, 20, 200),
                                     (36, 3, 64, 150), (20, 7, 300, 120)])
def test_native_plan_follows_the_oracle_and_the_generic_plan(env, C, K, F, N,
                                                             arith):
    zs, torch, dev = env
    if arith == 'bf16x3' and F > 256:
        pytest.skip('the bf16x3 kernels take <= 256 columns')
    X, y, w0 = _softmax_problem(C, K, F, N, seed=K * 100 + F)
    Xt, yt = torch.tensor(X, device=dev), torch.tensor(y, device=dev)

    def sampler(native):
        @zs.meta_bayesian_net()
        def model():
            bn = zs.BayesianNet()
            w = bn.normal('w', torch.zeros(K, F, device=dev), std=0.7,
                          n_samples=C, group_ndims=2)
            bn.categorical('y', Xt.unsqueeze(0) @ w.tensor.transpose(-1, -2),
                           group_ndims=1)
            return bn
        q = torch.tensor(w0, device=dev)
        hmc = zs.HMC(step_size=0.01, n_leapfrogs=5, adapt_step_size=True,
                     adapt_mass=True, mass_collect_iters=2, seed=21,
                     native_plans=native, likelihood_arithmetic=arith)
        op, info = hmc.sample(model(), {'y': yt}, {'w': q})
        return hmc, op, info, q

    hmc, op, info, q = sampler(True)
    assert hmc.plan_kind == 'linear_categorical', hmc.plan_reason
    assert hmc.likelihood_arithmetic_used == arith
    hg, opg, infog, qg = sampler(False)
    assert hg.plan_kind == 'generic'
    lj, grad = _oracle_model(X, y, 0.7)
    qr = [w0.copy()]
    ref = hmc_ref.HMC(step_size=0.01, n_leapfrogs=5, adapt_step_size=True,
                      adapt_mass=True, mass_collect_iters=2, seed=21)
    ref.sample(lj, grad, qr)
    n_flip = 0
    for it in range(6):
        op.run()
        opg.run()
        rinfo = ref.step()
        for f in ('orig_hamiltonian', 'hamiltonian', 'orig_log_prob'):
            # (the proposal's energy reaches 2e4 in the start-up transient,
            # where eps jumps to ~1: float32 relative, a few ulps)
            h = np.abs(getattr(rinfo, f)).max()
            np.testing.assert_allclose(getattr(info, f).cpu().numpy(),
                                       getattr(rinfo, f), rtol=0,
                                       atol=3e-5 * h + 1e-3)
            np.testing.assert_allclose(getattr(infog, f).cpu().numpy(),
                                       getattr(rinfo, f), rtol=0,
                                       atol=3e-5 * h + 1e-3)
        np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(),
                                   rinfo.acceptance_rate, atol=5e-3)
        np.testing.assert_allclose(float(info.updated_step_size.item()),
                                   float(ref.step_size), rtol=2e-3)
        got = q.cpu().numpy()
        bad = np.abs(got - qr[0]).reshape(C, -1).max(1) > 2e-3 * (
            1 + np.abs(qr[0]).max())
        n_flip += int(bad.sum())
        # teacher forcing: all three continue from the oracle's state
        q.copy_(torch.tensor(qr[0], device=dev))
        qg.copy_(torch.tensor(qr[0], device=dev))
    assert n_flip <= max(1, C * 6 // 100)
