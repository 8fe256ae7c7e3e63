"""zhusuan_amd/_symbolic.py on CPU tensors: the reference's literal dense
spellings (univariate.py:398-403 with logits = matmul(w, X^T);
lntm_mcem.py:39-46) stay symbolic until a distribution lowers them to the
fused likelihood's lazy operands; every other op computes on the plain
tensors exactly what it would without the wrapper."""
import numpy as np
import pytest
import torch

import zhusuan_amd as zs
from zhusuan_amd import _symbolic as sym


def _latent(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def test_literal_linear_logits_lower_to_the_fused_operand():
    w, X = _latent(6, 8), _latent(40, 8, seed=1)
    s = sym.wrap_latent(w)
    for logits in (s @ X.t(), torch.matmul(s, X.t()), s.matmul(X.t()),
                   torch.mm(s, X.t()), torch.nn.functional.linear(s, X)):
        assert isinstance(logits, sym.Sym) and logits.shape == (6, 40)
        lazy = sym.lower_bernoulli_logits(logits)
        assert isinstance(lazy, zs.distributions.LinearLogits)
        assert lazy.w is w
        assert lazy.X.data_ptr() == X.data_ptr() and lazy.X.shape == X.shape
        assert lazy.X.is_contiguous()
        torch.testing.assert_close(logits.force(), w @ X.t())
    d = zs.distributions.Bernoulli(s @ X.t(), group_ndims=1)
    assert d._lazy is not None and d._lazy.w is w
    assert tuple(d.get_batch_shape()) == (6, 40)


def test_literal_topic_model_logits_lower_to_the_fused_operand():
    n, n_docs, K, V = 3, 4, 8, 20
    eta, phi = _latent(n, n_docs, K), torch.softmax(_latent(K, V, seed=2), -1)
    s = sym.wrap_latent(eta)
    theta = torch.softmax(s, -1)                        # lntm_mcem.py:39
    pred = (theta.reshape(-1, K) @ phi).reshape(n, n_docs, V)   # :40-45
    logits = torch.log(pred)                                     # :46
    assert isinstance(logits, sym.Sym) and logits.shape == (n, n_docs, V)
    lazy = sym.lower_multinomial_logits(logits)
    assert isinstance(lazy, zs.distributions.LogMixture)
    assert lazy.softmax_source is eta and lazy.phi is phi
    assert tuple(lazy.shape) == (n, n_docs, V) and lazy._theta is None
    torch.testing.assert_close(lazy.theta, torch.softmax(eta, -1))
    torch.testing.assert_close(
        logits.force(), torch.log(torch.softmax(eta, -1) @ phi))
    # without the reshapes, and through zs.log_mixture on a symbolic theta
    lazy2 = sym.lower_multinomial_logits(torch.log(torch.softmax(s, -1) @ phi))
    assert lazy2.softmax_source is eta
    lazy3 = zs.log_mixture(torch.softmax(s, -1), phi)
    assert lazy3.softmax_source is eta and tuple(lazy3.shape) == (n, n_docs, V)
    d = zs.distributions.UnnormalizedMultinomial(
        logits, normalize_logits=False, dtype=torch.float32)
    assert d._lazy is not None and d._lazy.softmax_source is eta
    # re-normalised logits need the dense tensor (multivariate.py:440-441)
    d = zs.distributions.UnnormalizedMultinomial(logits, dtype=torch.float32)
    assert d._lazy is None and not isinstance(d.logits, sym.Sym)


@pytest.mark.parametrize('near_miss', [
    lambda s, X, phi: torch.softmax(s * 1.0, -1) @ phi,        # scaled latent
    lambda s, X, phi: torch.softmax(s, 0) @ phi,               # other axis
    lambda s, X, phi: torch.log(torch.softmax(s, -1) @ phi + 1e-6),
    lambda s, X, phi: (s @ X.t()) * 2.0,
    lambda s, X, phi: s.t() @ _latent(6, 5, seed=9),
    lambda s, X, phi: (s ** 2).sum(-1) + torch.sin(s).mean(),
    lambda s, X, phi: torch.cat([s, s], 0)[1:4] @ X.t(),
    lambda s, X, phi: s[:, :4] @ X.t()[:4],
])
def test_everything_else_computes_on_the_plain_tensor(near_miss):
    w = _latent(6, 8).requires_grad_(True)
    X, phi = _latent(40, 8, seed=1), torch.softmax(_latent(8, 20, seed=2), -1)
    want = near_miss(w, X, phi)
    got = sym.force(near_miss(sym.wrap_latent(w), X, phi))
    assert not isinstance(got, sym.Sym)
    torch.testing.assert_close(got, want, rtol=0, atol=0)
    g_want, = torch.autograd.grad(want.sum(), w)
    g_got, = torch.autograd.grad(got.sum(), w)
    torch.testing.assert_close(g_got, g_want, rtol=0, atol=0)
    # a distribution handed a symbol it cannot lower sees the value
    assert not isinstance(sym.lower_bernoulli_logits(
        near_miss(sym.wrap_latent(w), X, phi)), sym.Sym)


def test_metadata_and_meta_latents():
    w = _latent(6, 8)
    s = sym.wrap_latent(w)
    assert s.shape == (6, 8) and s.dim() == 2 and s.dtype == torch.float32
    assert s.device == w.device and s.size(0) == 6 and not s.requires_grad
    assert sym.wrap_latent(s) is s and sym.force(s) is w
    # a META latent against real constants: nothing executes, shapes follow
    m = sym.wrap_latent(torch.empty(6, 8, device='meta'))
    X = _latent(40, 8, seed=1)
    lazy = sym.lower_bernoulli_logits(m @ X.t())
    assert lazy.w.is_meta and tuple(lazy.shape) == (6, 40)


def test_model_function_with_literal_spelling_builds_fused_nodes():
    n, N, D = 5, 30, 8
    X = _latent(N, D, seed=1)
    y = (torch.rand(N) < 0.5).float()

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D), std=1., n_samples=n, group_ndims=1)
        bn.bernoulli('y', w.tensor @ X.t(), group_ndims=1,
                     dtype=torch.float32)
        return bn
    w = _latent(n, D, seed=3)
    bn = blr().observe(w=sym.wrap_latent(w), y=y)
    assert bn.get('y').dist._lazy.w is w
    # the StochasticTensor itself in the expression (bn.py:178-193 mixin)
    @zs.meta_bayesian_net()
    def blr2():
        bn = zs.BayesianNet()
        w = bn.normal('w', torch.zeros(D), std=1., n_samples=n, group_ndims=1)
        bn.bernoulli('y', w @ X.t(), group_ndims=1, dtype=torch.float32)
        return bn
    bn = blr2().observe(w=sym.wrap_latent(w), y=y)
    assert bn.get('y').dist._lazy.w is w


def test_product_autograd_functions_keep_the_tape_through_a_symbol():
    """`Function.apply` hands its arguments to `forward` without
    `__torch_function__` dispatch: the product's ops force symbols first."""
    from zhusuan_amd import _ops

    class Twice(_ops._Function):
        @staticmethod
        def forward(ctx, x, k):
            assert type(x) is torch.Tensor
            return x * k

        @staticmethod
        def backward(ctx, g):
            return g * 2, None
    w = _latent(5).requires_grad_(True)
    y = Twice.apply(sym.wrap_latent(w), 2.0)
    g, = torch.autograd.grad(y.sum(), w)
    torch.testing.assert_close(g, torch.full((5,), 2.0))


def test_a_foreign_autograd_function_is_detected_not_silently_cut():
    class Foreign(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 3.0          # forces the symbol with autograd off

        @staticmethod
        def backward(ctx, g):
            return g * 3.0
    w = _latent(5).requires_grad_(True)
    with pytest.raises(sym.SymbolicCut):
        Foreign.apply(sym.wrap_latent(w))
    # the sampler's evaluation falls back to plain tensors (and stays there)
    hmc = zs.HMC(step_size=0.1)
    hmc._log_joint = lambda obs: Foreign.apply(obs['w']).sum(-1) * 0 - \
        0.5 * (Foreign.apply(obs['w']) ** 2).sum(-1)
    hmc._observed = {}
    q = _latent(4, 5).requires_grad_(True)
    lp = hmc._eval_log_joint(['w'], [q])
    g, = torch.autograd.grad(lp.sum(), q)
    torch.testing.assert_close(g, -9.0 * q.detach())
    assert hmc._symbolic_latents is False
    # without grad nothing can be cut: symbols may be forced anywhere
    with torch.no_grad():
        assert torch.equal(Foreign.apply(sym.wrap_latent(_latent(3))),
                           _latent(3) * 3.0)


def test_odd_calls_on_symbols_compute_on_the_plain_tensor():
    w = _latent(6, 8).requires_grad_(True)
    s = sym.wrap_latent(w)
    assert s.requires_grad is True          # the latent's, not the wrapper's
    assert sym.wrap_latent(_latent(2, 2)).requires_grad is False
    v = s.view(torch.int32)                 # a dtype, not a shape
    assert not isinstance(v, sym.Sym) and v.dtype == torch.int32
    assert torch.equal(s.t(), w.t()) and len(s) == 6
    assert s.is_contiguous() and float(s.sum()) == float(w.sum())
    r = torch.reshape(s, shape=(8, 6))      # keyword form: plain tensor
    assert not isinstance(r, sym.Sym) and r.shape == (8, 6)
    z = s.reshape(3, 16)                    # last axis changes: plain tensor
    assert not isinstance(z, sym.Sym)


def test_sums_of_linear_terms_and_a_bias_lower_to_one_fused_operand():
    """logits = w1 @ X1^T + w2 @ X2^T + b (every spelling of `+` and of the
    bias column) -> one LinearLogits whose terms are the latents themselves."""
    C, N = 6, 40
    w1, w2 = _latent(C, 8), _latent(C, 3, seed=3)
    b1, b0 = _latent(C, 1, seed=4), _latent(C, seed=5)
    X1, X2 = _latent(N, 8, seed=1), _latent(N, 3, seed=2)
    s1, s2, sb1, sb0 = (sym.wrap_latent(t) for t in (w1, w2, b1, b0))
    cases = [
        (s1 @ X1.t() + sb1, [(w1, X1, False), (b1, None, False)]),
        (sb1 + s1 @ X1.t(), [(b1, None, False), (w1, X1, False)]),
        (torch.add(s1 @ X1.t(), sb0[:, None]),
         [(w1, X1, False), (b0, None, True)]),
        ((s1 @ X1.t()).add(sb0.unsqueeze(-1)),
         [(w1, X1, False), (b0, None, True)]),
        (s1 @ X1.t() + sb0[..., None], [(w1, X1, False), (b0, None, True)]),
        (s1 @ X1.t() + s2 @ X2.t() + torch.unsqueeze(sb0, 1),
         [(w1, X1, False), (w2, X2, False), (b0, None, True)]),
    ]
    for logits, want in cases:
        assert isinstance(logits, sym.Sym) and logits.shape == (C, N)
        lazy = sym.lower_bernoulli_logits(logits)
        assert isinstance(lazy, zs.distributions.LinearLogits)
        assert len(lazy.terms) == len(want)
        for (w, X, sc), (w_w, X_w, sc_w) in zip(lazy.terms, want):
            assert w is w_w and sc == sc_w
            assert (X is None) == (X_w is None)
            if X is not None:
                assert X.data_ptr() == X_w.data_ptr() and X.shape == X_w.shape
        dense = sum((w.unsqueeze(-1) if sc else w) if X is None else w @ X.t()
                    for w, X, sc in want)
        torch.testing.assert_close(logits.force(), dense)
        torch.testing.assert_close(lazy.dense(), dense)
        assert lazy.n_rows == N and tuple(lazy.shape) == (C, N)
        assert lazy.n_features == sum(1 if X is None else X.shape[1]
                                      for _, X, _ in want)
    # the explicit form
    lazy = zs.linear_logits(w1, X1, bias=b0)
    assert lazy.terms[1][0] is b0 and lazy.terms[1][2] is True
    lazy = zs.linear_logits(w1, X1, bias=b1)
    assert lazy.terms[1][0] is b1 and lazy.terms[1][2] is False
    with pytest.raises(ValueError):
        zs.linear_logits(w1, X1, bias=_latent(C, 2))


@pytest.mark.parametrize('near_miss', [
    lambda s, b, X: s @ X.t() + 1.0,                   # constant offset
    lambda s, b, X: s @ X.t() + torch.ones(40),        # constant row
    lambda s, b, X: torch.add(s @ X.t(), b[:, None], alpha=2.0),
    lambda s, b, X: s @ X.t() + s @ X.t(),             # the same latent twice
    lambda s, b, X: s @ X.t() - b[:, None],
    lambda s, b, X: b[:, None] + b[:, None],           # no design matrix
    lambda s, b, X: s @ X.t() + b[None, :6].t(),       # not the bias column
])
def test_sums_outside_the_spelling_compute_on_the_plain_tensor(near_miss):
    w, b = _latent(6, 8), _latent(6, seed=4)
    X = _latent(40, 8, seed=1)
    want = near_miss(w, b, X)
    out = near_miss(sym.wrap_latent(w), sym.wrap_latent(b), X)
    got = sym.lower_bernoulli_logits(out) if isinstance(out, sym.Sym) else out
    assert not isinstance(got, (sym.Sym, zs.distributions.LinearLogits))
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_packed_operands_of_a_multi_term_linear_logits():
    """LinearLogits.packed(): the weights side by side (torch.cat, so autograd
    splits the gradient back), the design matrices side by side with a column
    of ones for the bias -- and the same product as the dense sum."""
    from zhusuan_amd import _ops
    C, N = 5, 17
    w1 = _latent(C, 6).requires_grad_(True)
    w2 = _latent(C, 3, seed=3).requires_grad_(True)
    b = _latent(C, seed=5).requires_grad_(True)
    X1, X2 = _latent(N, 6, seed=1), _latent(N, 3, seed=2)
    lazy = zs.distributions.LinearLogits.of_terms(
        [(w1, X1, False), (b, None, True), (w2, X2, False)])
    w_all, x_all = lazy.packed()
    assert tuple(w_all.shape) == (C, 10) and tuple(x_all.shape) == (N, 10)
    torch.testing.assert_close(x_all[:, :6], X1)
    torch.testing.assert_close(x_all[:, 6], torch.ones(N))
    torch.testing.assert_close(x_all[:, 7:], X2)
    dense = w1 @ X1.t() + b[:, None] + w2 @ X2.t()
    torch.testing.assert_close(w_all @ x_all.t(), dense)
    torch.testing.assert_close(lazy.dense(), dense)
    (w_all @ x_all.t()).sum().backward()
    torch.testing.assert_close(w1.grad, X1.sum(0).expand(C, 6))
    torch.testing.assert_close(b.grad, torch.full((C,), float(N)))
    # the cached design matrix is rebuilt when a block changes in place
    again = _ops.packed_design([X1, None, X2], N, X1.device)
    assert again.data_ptr() == x_all.data_ptr()
    X2.mul_(2.0)
    fresh = _ops.packed_design([X1, None, X2], N, X1.device)
    torch.testing.assert_close(fresh[:, 7:], X2)
    # zero padding to a kernel width
    wide = _ops.packed_design([X1, None], N, X1.device, 64)
    assert tuple(wide.shape) == (N, 64) and not wide[:, 7:].any()
    _ops.clear_caches()
    # inconsistent terms are refused
    with pytest.raises(ValueError):
        zs.distributions.LinearLogits.of_terms(
            [(w1, X1, False), (w2, _latent(N + 1, 3), False)])
    with pytest.raises(ValueError):
        zs.distributions.LinearLogits.of_terms(
            [(w1, X1, False), (_latent(C + 1, 3), X2, False)])
