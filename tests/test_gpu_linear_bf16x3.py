"""GPU parity of the bf16x3 likelihood kernels (csrc/b3_kernel.h: three
bfloat16 planes per float32 operand, six bf16 MFMAs per product, float32
accumulation) against a float64 restatement of Bernoulli._log_prob
(reference zhusuan/distributions/univariate.py:398-403) /
UnnormalizedMultinomial._log_prob (multivariate.py:435-443) on materialised
logits and of the gradient tf.gradients (hmc.py:430-432) gives through them
-- at the tolerances the exact-fp32 kernels are held to
(tests/test_gpu_linear_bernoulli.py, test_gpu_mixture_multinomial.py) -- and
against the fp32 kernels themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from zhusuan_amd import _capi
    assert torch.cuda.is_available()
    return torch, _capi, torch.device('cuda', 0)


def _image(torch, _capi, X, width):
    """X [N, width] float32 on the device -> the kernel's tile image."""
    import ctypes
    nb = ctypes.c_int64()
    _capi.call('zshmc_bf16x3_image_bytes', X.shape[0], width,
               ctypes.addressof(nb))
    img = torch.empty(nb.value, dtype=torch.uint8, device=X.device)
    _capi.call('zshmc_bf16x3_split', X.data_ptr(), X.shape[0], width,
               X.stride(0), img.data_ptr(), _capi.current_stream())
    return img


def _bf16_planes(x):
    """float64 emulation of the kernel's split: (hi, mid, lo) as float32."""
    def rne(v):
        u = np.asarray(v, np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)
    hi = rne(x)
    mid = rne(x - hi)
    lo = rne(x - hi - mid)
    return hi, mid, lo


def test_split_image_matches_the_layout_in_the_header(env):
    """zshmc_bf16x3_split against a NumPy restatement of the image layout
    (include/zshmc.h, csrc/b3_kernel.h: b3_chunk): bit for bit."""
    torch, _capi, dev = env
    rng = np.random.RandomState(0)
    N, D = 77, 128
    X = rng.normal(size=(N, D)).astype(np.float32)
    X[5, 3] = 0.0
    X[6, :] *= 1e-20
    img = _image(torch, _capi, torch.tensor(X, device=dev), D).cpu().numpy()
    planes = _bf16_planes(X)
    nt, np_ = (N + 31) // 32, D // 32
    want = np.zeros((nt, 3, np_, 128, 8), np.uint16)
    for p in range(3):
        b = (planes[p].view(np.uint32) >> 16).astype(np.uint16)
        for n in range(N):
            t, m = divmod(n, 32)
            for c8 in range(D // 8):
                P, par, h = c8 >> 2, (c8 >> 1) & 1, c8 & 1
                chunk = (m >> 2) * 16 + 4 * ((2 * h + par + (m >> 3)) & 3) + (m & 3)
                want[t, p, P, chunk] = b[n, c8 * 8:c8 * 8 + 8]
    got = img.view(np.uint16).reshape(nt, 3, np_, 128, 8)
    np.testing.assert_array_equal(got, want)
    # and the three planes add up to the value exactly
    np.testing.assert_array_equal(
        (planes[0].astype(np.float64) + planes[1] + planes[2]
         ).astype(np.float32), X)


def _bern_data(C, N, D, seed):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(N, D)).astype(np.float32)
    w_true = rng.normal(size=D).astype(np.float32)
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true / np.sqrt(D)))
         ).astype(np.float32)
    W = (rng.normal(size=(C, D)) * 0.3).astype(np.float32)
    return X, y, W


def _bern_ref(W, X, y):
    l = W.astype(np.float64) @ X.astype(np.float64).T
    ll = (y * l - np.maximum(l, 0) - np.log1p(np.exp(-np.abs(l)))).sum(1)
    g = (y - 1 / (1 + np.exp(-l))) @ X.astype(np.float64)
    return ll, g


def _call_bern(torch, _capi, dev, W, X, y, D, want_ll, n_splits=1, fp32=False):
    C, N = W.shape[0], X.shape[0]
    Wt, Xt, yt = (torch.tensor(a, device=dev) for a in (W, X, y))
    ll = torch.full((C,), float('nan'), device=dev) if want_ll else None
    g = torch.full((C, D), float('nan'), device=dev)
    ws = torch.empty(n_splits * C * (D + 1), device=dev) if n_splits > 1 else None
    s = _capi.current_stream()
    if fp32:
        _capi.call('zshmc_linear_bernoulli_log_lik', Wt.data_ptr(),
                   Xt.data_ptr(), yt.data_ptr(), C, N, D, _capi.ptr(ll),
                   g.data_ptr(), n_splits, _capi.ptr(ws), s)
    else:
        img = _image(torch, _capi, Xt, D)
        _capi.call('zshmc_linear_bernoulli_log_lik_bf16x3', Wt.data_ptr(),
                   img.data_ptr(), yt.data_ptr(), C, N, D, _capi.ptr(ll),
                   g.data_ptr(), n_splits, _capi.ptr(ws), s)
    torch.cuda.synchronize()
    return (ll.cpu().numpy() if want_ll else None), g.cpu().numpy()


# every width; ragged chain blocks (128), ragged 32-row tiles, one tile, one
# row, several tiles; chains past a multiple of 128
@pytest.mark.parametrize('C,N,D', [
    (128, 32, 64), (128, 64, 128), (100, 1000, 256), (7, 45, 128),
    (130, 333, 64), (64, 4096, 192), (1, 1, 64), (256, 10000, 256),
    (300, 777, 128), (3, 130, 192), (129, 31, 256), (128, 33, 256)])
@pytest.mark.parametrize('want_ll', [True, False])
def test_bernoulli_matches_float64_reference(env, C, N, D, want_ll):
    torch, _capi, dev = env
    X, y, W = _bern_data(C, N, D, seed=C + N + D)
    ll, g = _call_bern(torch, _capi, dev, W, X, y, D, want_ll)
    ll_ref, g_ref = _bern_ref(W, X, y)
    if want_ll:
        np.testing.assert_allclose(ll, ll_ref, rtol=2e-5, atol=2e-5 * N)
    scale = np.abs(g_ref).max() + 1.0
    np.testing.assert_allclose(g, g_ref, rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize('D', [64, 128, 192, 256])
def test_bernoulli_is_as_close_to_float64_as_the_fp32_kernel(env, D):
    """The claim of the design (tools/bf16x3_accuracy.py on the device): the
    six-term split leaves the error of float32 accumulation itself -- within
    2x of what the exact-fp32 MFMA kernel leaves on the same inputs."""
    torch, _capi, dev = env
    C, N = 256, 20000
    X, y, W = _bern_data(C, N, D, seed=D)
    ll_ref, g_ref = _bern_ref(W, X, y)
    ll3, g3 = _call_bern(torch, _capi, dev, W, X, y, D, True)
    ll1, g1 = _call_bern(torch, _capi, dev, W, X, y, D, True, fp32=True)
    e3 = np.abs(g3 - g_ref).max() / np.abs(g_ref).max()
    e1 = np.abs(g1 - g_ref).max() / np.abs(g_ref).max()
    l3 = np.abs(ll3 - ll_ref).max()
    l1 = np.abs(ll1 - ll_ref).max()
    print('D=%d gradient max rel err: bf16x3 %.2e fp32 %.2e; log-lik max abs '
          'err: bf16x3 %.2e fp32 %.2e (of %.0f)' % (
              D, e3, e1, l3, l1, np.abs(ll_ref).max()))
    assert e3 < 2 * e1 + 1e-7
    assert l3 < 2 * l1 + 1e-3


def test_bernoulli_row_splits_and_bit_stability(env):
    torch, _capi, dev = env
    C, N, D = 100, 5000, 128
    X, y, W = _bern_data(C, N, D, seed=3)
    ll_ref, g_ref = _bern_ref(W, X, y)
    ll_a, g_a = _call_bern(torch, _capi, dev, W, X, y, D, True, n_splits=5)
    ll_b, g_b = _call_bern(torch, _capi, dev, W, X, y, D, True, n_splits=5)
    np.testing.assert_array_equal(ll_a, ll_b)
    np.testing.assert_array_equal(g_a, g_b)
    np.testing.assert_allclose(ll_a, ll_ref, rtol=2e-5, atol=2e-5 * N)
    np.testing.assert_allclose(g_a, g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))
    # more slices than tiles
    ll_c, g_c = _call_bern(torch, _capi, dev, W[:, :64].copy(), X[:40, :64].copy(),
                           y[:40], 64, True, n_splits=4)
    ll_r, g_r = _bern_ref(W[:, :64], X[:40, :64], y[:40])
    np.testing.assert_allclose(ll_c, ll_r, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(g_c, g_r, rtol=1e-4, atol=1e-3)


def test_bernoulli_nonfinite_chain_stays_in_its_rows(env):
    """A diverged chain (inf / nan weights) poisons its own results only."""
    torch, _capi, dev = env
    C, N, D = 128, 200, 64
    X, y, W = _bern_data(C, N, D, seed=9)
    W[5, 3] = np.inf
    W[70, 0] = np.nan
    ll, g = _call_bern(torch, _capi, dev, W, X, y, D, True)
    ok = np.ones(C, bool)
    ok[[5, 70]] = False
    Wc = W.copy()
    Wc[~ok] = 0
    ll_ref, g_ref = _bern_ref(Wc, X, y)
    assert np.isfinite(ll[ok]).all() and np.isfinite(g[ok]).all()
    assert not np.isfinite(ll[~ok]).any()
    np.testing.assert_allclose(ll[ok], ll_ref[ok], rtol=2e-5, atol=2e-5 * N)
    np.testing.assert_allclose(g[ok], g_ref[ok], rtol=1e-4, atol=1e-3)


def _mult_data(n_chains, n_docs, V, K, seed):
    rng = np.random.RandomState(seed)
    eta = rng.normal(size=(n_chains * n_docs, K))
    theta = np.exp(eta - eta.max(1, keepdims=True))
    theta = (theta / theta.sum(1, keepdims=True)).astype(np.float32)
    beta = rng.normal(size=(K, V))
    phi = np.exp(beta - beta.max(1, keepdims=True))
    phi = (phi / phi.sum(1, keepdims=True)).astype(np.float32)
    x = rng.poisson(0.3, size=(n_docs, V)).astype(np.float32)
    return theta, phi, x


def _mult_ref(theta, phi, x, n_docs):
    S = theta.astype(np.float64) @ phi.astype(np.float64)       # [R, V]
    R = theta.shape[0]
    xr = x[np.arange(R) % n_docs].astype(np.float64)            # row r: doc r % n_docs
    with np.errstate(divide='ignore', invalid='ignore'):
        ll = np.where(xr != 0, xr * np.log(S), 0.0).sum(1)
        g = np.where(xr != 0, xr / S, 0.0) @ phi.astype(np.float64).T
    return ll, g


@pytest.mark.parametrize('n_chains,n_docs,V,K', [
    (128, 3, 500, 64), (130, 2, 333, 128), (64, 1, 1000, 128),
    (5, 7, 77, 256), (256, 2, 1241, 128), (40, 3, 100, 192)])
@pytest.mark.parametrize('want_ll', [True, False])
def test_multinomial_matches_float64_reference(env, n_chains, n_docs, V, K,
                                               want_ll):
    torch, _capi, dev = env
    theta, phi, x = _mult_data(n_chains, n_docs, V, K, seed=V + K)
    R = n_chains * n_docs
    stride = (V + 3) // 4 * 4
    xp = np.zeros((n_docs, stride), np.float32)
    xp[:, :V] = x
    th = torch.tensor(theta, device=dev)
    pt = torch.tensor(np.ascontiguousarray(phi.T), device=dev)   # [V, K]
    xt = torch.tensor(xp, device=dev)
    img = _image(torch, _capi, pt, K)
    ll = torch.full((R,), float('nan'), device=dev) if want_ll else None
    g = torch.full((R, K), float('nan'), device=dev)
    _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', th.data_ptr(),
               img.data_ptr(), xt.data_ptr(), n_docs, stride, R, V, K,
               _capi.ptr(ll), g.data_ptr(), 1, None, _capi.current_stream())
    torch.cuda.synchronize()
    ll_ref, g_ref = _mult_ref(theta, phi, x, n_docs)
    if want_ll:
        np.testing.assert_allclose(ll.cpu().numpy(), ll_ref, rtol=2e-5,
                                   atol=2e-5 * V)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))


@pytest.mark.parametrize('n_chains,n_docs,V,K,n_splits', [
    (1, 100, 1241, 128, 1), (1, 100, 1241, 128, 7), (3, 50, 333, 64, 1),
    (2, 77, 500, 192, 3), (5, 7, 77, 128, 1), (1, 300, 1000, 128, 2),
    (127, 3, 96, 64, 1), (1, 2, 31, 64, 1)])
@pytest.mark.parametrize('want_ll', [True, False])
def test_multinomial_packed_rows_match_float64_reference(env, n_chains, n_docs,
                                                         V, K, n_splits,
                                                         want_ll):
    """ABI 0.5.1: a few chains x many documents (lntm_mcem.py:62-70 runs
    n_chains = 1) -- a workgroup takes 128 CONSECUTIVE (chain, document) rows,
    every row with its own counts row, brought tile by tile as [32 vocabulary
    rows][128 chains] by four 16-byte-per-lane DMAs per wave.  Taken when the
    counts rows are padded to 32 floats; ragged last blocks, vocabularies that
    are not whole tiles, vocabulary slices."""
    torch, _capi, dev = env
    lib = _capi.load()
    assert lib.zshmc_bf16x3_multinomial_rows_packed(n_docs, n_chains) == 1
    assert lib.zshmc_bf16x3_multinomial_rows_packed(n_docs, 128) == 0
    assert lib.zshmc_bf16x3_multinomial_rows_packed(n_docs, 1024) == 0
    assert lib.zshmc_bf16x3_multinomial_rows_packed(1, n_chains) == 0
    theta, phi, x = _mult_data(n_chains, n_docs, V, K, seed=V + K + n_docs)
    R = n_chains * n_docs
    stride = (V + 31) // 32 * 32
    xp = np.zeros((n_docs, stride), np.float32)
    xp[:, :V] = x
    th = torch.tensor(theta, device=dev)
    pt = torch.tensor(np.ascontiguousarray(phi.T), device=dev)   # [V, K]
    xt = torch.tensor(xp, device=dev)
    img = _image(torch, _capi, pt, K)
    ws = torch.empty(n_splits * R * (K + 1), device=dev) if n_splits > 1 \
        else None
    out = []
    for rep in range(2):
        ll = torch.full((R,), float('nan'), device=dev) if want_ll else None
        g = torch.full((R, K), float('nan'), device=dev)
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', th.data_ptr(),
                   img.data_ptr(), xt.data_ptr(), n_docs, stride, R, V, K,
                   _capi.ptr(ll), g.data_ptr(), n_splits, _capi.ptr(ws),
                   _capi.current_stream())
        torch.cuda.synchronize()
        out.append((None if ll is None else ll.cpu().numpy(),
                    g.cpu().numpy()))
    np.testing.assert_array_equal(out[0][1], out[1][1])        # bit-stable
    ll_ref, g_ref = _mult_ref(theta, phi, x, n_docs)
    if want_ll:
        np.testing.assert_array_equal(out[0][0], out[1][0])
        np.testing.assert_allclose(out[0][0], ll_ref, rtol=2e-5,
                                   atol=2e-5 * V)
    np.testing.assert_allclose(out[0][1], g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))
    # ... and the same rows through the one-document-per-workgroup form (counts
    # rows padded to 4 floats only: the packed form's precondition fails)
    s4 = (V + 3) // 4 * 4
    if s4 != stride:
        x4 = torch.tensor(np.ascontiguousarray(xp[:, :s4]), device=dev)
        g4 = torch.full((R, K), float('nan'), device=dev)
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', th.data_ptr(),
                   img.data_ptr(), x4.data_ptr(), n_docs, s4, R, V, K, None,
                   g4.data_ptr(), n_splits, _capi.ptr(ws),
                   _capi.current_stream())
        torch.cuda.synchronize()
        np.testing.assert_allclose(g4.cpu().numpy(), out[0][1], rtol=2e-5,
                                   atol=2e-6 * (np.abs(g_ref).max() + 1))


@pytest.mark.parametrize('n_chains,n_docs,V,K,n_splits,rate', [
    (128, 3, 500, 64, 1, 0.3), (130, 2, 333, 128, 1, 0.05),
    (64, 1, 1000, 128, 1, 0.1), (256, 2, 1241, 128, 3, 0.08),
    (40, 3, 100, 192, 1, 0.3), (5, 7, 77, 256, 1, 0.5),
    (128, 4, 12419, 128, 4, 0.08), (256, 3, 2000, 256, 2, 0.02)])
@pytest.mark.parametrize('want_ll', [True, False])
def test_multinomial_own_vocabulary_matches_float64_reference(
        env, n_chains, n_docs, V, K, n_splits, rate, want_ll):
    """ABI 0.6.0: one document per workgroup, the tile loop over the
    document's OWN vocabulary (words with a zero count contribute exactly
    nothing): tiles gathered from the phi^T image through per-lane DMA
    offsets.  Documents without words, documents that use every word, word
    lists that are not whole tiles, row-range slices -- against float64 and
    against the dense form of the same kernel."""
    torch, _capi, dev = env
    from zhusuan_amd import _ops
    theta, phi, _ = _mult_data(n_chains, n_docs, V, K, seed=V + K + n_docs)
    rng = np.random.RandomState(V + n_docs)
    x = rng.poisson(rate, size=(n_docs, V)).astype(np.float32)
    if n_docs > 1:
        x[-1] = 0.0                      # a document without words
    if n_docs > 2:
        x[1] = 1.0 + rng.poisson(1.0, size=V)     # ... and one with all of them
    R = n_chains * n_docs
    th = torch.tensor(theta, device=dev)
    pt = torch.tensor(np.ascontiguousarray(phi.T), device=dev)   # [V, K]
    xt = torch.tensor(x, device=dev)
    img = _image(torch, _capi, pt, K)
    vals, rows, off, total = _ops.counts_csr(xt)
    assert total % 32 == 0 and off.shape[0] == n_docs + 1
    lens = (off[1:] - off[:-1]).cpu().numpy()
    assert (lens % 32 == 0).all() and (lens >= 32).all()
    assert int(rows.max()) < V and float(vals.sum()) == float(x.sum())
    ws = torch.empty(n_splits * R * (K + 1), device=dev) if n_splits > 1 \
        else None
    out = []
    for rep in range(2):
        ll = torch.full((R,), float('nan'), device=dev) if want_ll else None
        g = torch.full((R, K), float('nan'), device=dev)
        _capi.call('zshmc_linear_multinomial_log_lik_bf16x3_sparse',
                   th.data_ptr(), img.data_ptr(), vals.data_ptr(),
                   rows.data_ptr(), off.data_ptr(), n_docs, R, V, K,
                   _capi.ptr(ll), g.data_ptr(), n_splits, _capi.ptr(ws),
                   _capi.current_stream())
        torch.cuda.synchronize()
        out.append((None if ll is None else ll.cpu().numpy(),
                    g.cpu().numpy()))
    np.testing.assert_array_equal(out[0][1], out[1][1])        # bit-stable
    ll_ref, g_ref = _mult_ref(theta, phi, x, n_docs)
    if want_ll:
        np.testing.assert_array_equal(out[0][0], out[1][0])
        np.testing.assert_allclose(out[0][0], ll_ref, rtol=2e-5,
                                   atol=2e-5 * V)
    np.testing.assert_allclose(out[0][1], g_ref, rtol=1e-4,
                               atol=2e-5 * (np.abs(g_ref).max() + 1))
    # the dense form of the same kernel on the same operands
    s4 = (V + 3) // 4 * 4
    xp = np.zeros((n_docs, s4), np.float32)
    xp[:, :V] = x
    x4 = torch.tensor(xp, device=dev)
    g4 = torch.full((R, K), float('nan'), device=dev)
    _capi.call('zshmc_linear_multinomial_log_lik_bf16x3', th.data_ptr(),
               img.data_ptr(), x4.data_ptr(), n_docs, s4, R, V, K, None,
               g4.data_ptr(), n_splits, _capi.ptr(ws), _capi.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0][1], g4.cpu().numpy(), rtol=2e-5,
                               atol=2e-6 * (np.abs(g_ref).max() + 1))


# ---- Categorical (OP 2): rows of W are (chain, class) pairs ----------------
def _cat_ref(w, X, y):
    """ll [C], d ll / d w [C, K, F] in float64 (univariate.py:496-548 on
    materialised logits; hmc.py:430-432)."""
    w, X = w.astype(np.float64), X.astype(np.float64)
    logits = np.einsum('nf,ckf->cnk', X, w)
    m = logits.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(-1))
    ll = (np.take_along_axis(logits, y[None, :, None].astype(np.int64),
                             -1)[..., 0] - lse).sum(-1)
    res = -np.exp(logits - lse[..., None])
    res[:, np.arange(len(y)), y] += 1.0
    return ll, np.einsum('cnk,nf->ckf', res, X)


def _call_cat(torch, _capi, dev, w, X, y, G, D, want_ll, n_splits=1,
              fp32=False):
    """w [C, K, F] -> the kernel's operand [C * G, D] (padding classes and
    columns zero); returns ll [C, G], grad [C, G, D]."""
    C, K, F = w.shape
    N = X.shape[0]
    Wp = np.zeros((C, G, D), np.float32)
    Wp[:, :K, :F] = w
    Xp = np.zeros((N, D), np.float32)
    Xp[:, :F] = X
    Wt, Xt = torch.tensor(Wp, device=dev), torch.tensor(Xp, device=dev)
    yt = torch.tensor(y.astype(np.float32), device=dev)
    ll = torch.full((C * G,), float('nan'), device=dev) if want_ll else None
    g = torch.full((C * G, D), float('nan'), device=dev)
    ws = torch.empty(n_splits * C * G * (D + 1), device=dev) \
        if n_splits > 1 else None
    s = _capi.current_stream()
    if fp32:
        _capi.call('zshmc_linear_categorical_log_lik', Wt.data_ptr(),
                   Xt.data_ptr(), yt.data_ptr(), C * G, N, D, K, G,
                   _capi.ptr(ll), g.data_ptr(), n_splits, _capi.ptr(ws), s)
    else:
        img = _image(torch, _capi, Xt, D)
        _capi.call('zshmc_linear_categorical_log_lik_bf16x3', Wt.data_ptr(),
                   img.data_ptr(), yt.data_ptr(), C * G, N, D, K, G,
                   _capi.ptr(ll), g.data_ptr(), n_splits, _capi.ptr(ws), s)
    torch.cuda.synchronize()
    return (ll.cpu().numpy().reshape(C, G) if want_ll else None), \
        g.cpu().numpy().reshape(C, G, D)


def _cat_data(C, K, F, N, seed):
    rng = np.random.RandomState(seed)
    X = rng.normal(size=(N, F)).astype(np.float32)
    w = (rng.normal(size=(C, K, F)) / np.sqrt(F)).astype(np.float32)
    w[0] *= 30.0                       # one chain with |logits| ~ 30
    y = rng.randint(0, K, size=N).astype(np.int32)
    return X, y, w


# every class stride 1 .. 32 with and without padding classes, every width,
# ragged 128-row blocks and 32-row tiles
@pytest.mark.parametrize('C,K,G,F,D,N', [
    (10, 4, 4, 5, 64, 30),        # the reference trace's shape
    (70, 2, 2, 64, 64, 130),
    (33, 3, 4, 17, 64, 257),      # a padding class, padding columns
    (40, 10, 16, 100, 128, 515),  # MNIST-like class count
    (24, 16, 16, 128, 128, 300),
    (20, 8, 8, 150, 192, 333),
    (9, 32, 32, 256, 256, 96),    # a full half-wave of classes
    (5, 20, 32, 200, 256, 1000),  # 12 padding classes
    (130, 5, 8, 8, 64, 64),
    (300, 1, 1, 30, 64, 50),      # one class: log-lik 0, gradient 0
])
@pytest.mark.parametrize('want_ll', [True, False])
def test_categorical_matches_float64_reference(env, C, K, G, F, D, N, want_ll):
    torch, _capi, dev = env
    X, y, w = _cat_data(C, K, F, N, seed=C * 1000 + K)
    ll, g = _call_cat(torch, _capi, dev, w, X, y, G, D, want_ll)
    ll_ref, g_ref = _cat_ref(w, X, y)
    if want_ll:
        # (tests/test_gpu_linear_categorical.py's tolerances)
        scale = np.abs(ll_ref).max()
        np.testing.assert_allclose(ll.sum(-1), ll_ref, rtol=2e-6,
                                   atol=2e-6 * scale + 1e-4)
        assert np.abs(ll[:, K:]).max(initial=0.0) == 0.0
    gs = np.abs(g_ref).max()
    np.testing.assert_allclose(g[:, :K, :F], g_ref, rtol=0,
                               atol=2e-5 * gs + 1e-5)
    # padding classes and padding columns: exact zeros
    assert np.abs(g[:, K:]).max(initial=0.0) == 0.0
    assert np.abs(g[:, :, F:]).max(initial=0.0) == 0.0


def test_categorical_is_as_close_to_float64_as_the_fp32_kernel(env):
    torch, _capi, dev = env
    C, K, G, F, D, N = 64, 10, 16, 128, 128, 20000
    X, y, w = _cat_data(C, K, F, N, seed=7)
    ll_ref, g_ref = _cat_ref(w, X, y)
    ll3, g3 = _call_cat(torch, _capi, dev, w, X, y, G, D, True)
    ll1, g1 = _call_cat(torch, _capi, dev, w, X, y, G, D, True, fp32=True)
    e3 = np.abs(g3[:, :K] - g_ref).max() / np.abs(g_ref).max()
    e1 = np.abs(g1[:, :K] - g_ref).max() / np.abs(g_ref).max()
    l3 = np.abs(ll3.sum(-1) - ll_ref).max()
    l1 = np.abs(ll1.sum(-1) - ll_ref).max()
    print('categorical gradient max rel err: bf16x3 %.2e fp32 %.2e; log-lik '
          'max abs err: bf16x3 %.2e fp32 %.2e (of %.0f)' % (
              e3, e1, l3, l1, np.abs(ll_ref).max()))
    assert e3 < 2 * e1 + 1e-7
    assert l3 < 2 * l1 + 1e-3


def test_categorical_row_splits_and_bit_stability(env):
    torch, _capi, dev = env
    C, K, G, F, D, N = 8, 10, 16, 100, 128, 6000
    X, y, w = _cat_data(C, K, F, N, seed=1)
    ll_ref, g_ref = _cat_ref(w, X, y)
    a = _call_cat(torch, _capi, dev, w, X, y, G, D, True, n_splits=4)
    b = _call_cat(torch, _capi, dev, w, X, y, G, D, True, n_splits=4)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[0].sum(-1), ll_ref, rtol=2e-6,
                               atol=2e-6 * np.abs(ll_ref).max() + 1e-4)
    np.testing.assert_allclose(a[1][:, :K, :F], g_ref, rtol=0,
                               atol=2e-5 * np.abs(g_ref).max() + 1e-5)


def test_categorical_refuses_what_the_kernel_does_not_take(env):
    torch, _capi, dev = env
    t = torch.zeros(128 * 512, device=dev)
    img = torch.zeros(6 * 32 * 512, dtype=torch.uint8, device=dev)

    def call(n_cols, n_rows, D, K, G, grad=True):
        _capi.call('zshmc_linear_categorical_log_lik_bf16x3', t.data_ptr(),
                   img.data_ptr(), t.data_ptr(), n_cols, n_rows, D, K, G,
                   t.data_ptr(), t.data_ptr() if grad else None, 1, None,
                   _capi.current_stream())
    with pytest.raises(_capi.ZshmcError, match='power of two'):
        call(12, 4, 64, 3, 3)
    with pytest.raises(_capi.ZshmcError, match='power of two'):
        call(64, 4, 64, 40, 64)
    with pytest.raises(_capi.ZshmcError, match='bad shape'):
        call(10, 4, 64, 4, 4)          # columns not whole chains
    with pytest.raises(_capi.ZshmcError, match='bad shape'):
        call(16, 4, 320, 4, 4)         # a width of the fp32 kernels only
    with pytest.raises(_capi.ZshmcError, match='null pointer'):
        call(16, 4, 64, 4, 4, grad=False)


def test_hmc_on_bf16x3_run_many_equals_a_loop_of_runs(env):
    """HMC(likelihood_arithmetic='bf16x3') through both launch loops -- the
    Python one (sample_op.run) and the C one (zshmc_hmc_model_run:
    sample_op.run_many) -- bit for bit, with the start evaluation carried and
    with every start evaluated (reuse_start_evaluation=False)."""
    torch, _capi, dev = env
    import zhusuan_amd as zs
    rng = np.random.RandomState(4)
    N, D, C = 500, 40, 256
    X = torch.tensor(rng.normal(size=(N, D)).astype(np.float32), device=dev)
    y = torch.tensor((rng.uniform(size=N) < 0.5).astype(np.int32), device=dev)
    w0 = (0.1 * rng.normal(size=(C, D))).astype(np.float32)

    @zs.meta_bayesian_net()
    def blr():
        bn = zs.BayesianNet()
        wn = bn.normal('w', torch.zeros(D, device=dev), std=1., n_samples=C,
                       group_ndims=1)
        bn.bernoulli('y', wn.tensor @ X.t(), group_ndims=1)
        return bn
    out = {}
    for mode in ('loop', 'block', 'loop_evaluate', 'block_evaluate'):
        flag = zs.placeholder(bool)
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=5, seed=11,
                     adapt_step_size=flag, likelihood_arithmetic='bf16x3',
                     reuse_start_evaluation=not mode.endswith('evaluate'))
        w = torch.tensor(w0, device=dev)
        op, info = hmc.sample(blr(), {'y': y}, {'w': w})
        assert hmc.plan_kind == 'linear_bernoulli'
        assert hmc.likelihood_arithmetic_used == 'bf16x3'
        for n, feed in ((5, {flag: True}), (4, {flag: False})):
            if mode.startswith('block'):
                op.run_many(n, feed_dict=feed)
            else:
                for _ in range(n):
                    op.run(feed_dict=feed)
        out[mode] = (w.cpu().numpy(), info.log_prob.cpu().numpy(),
                     float(info.updated_step_size.item()))
    for mode in ('block', 'loop_evaluate', 'block_evaluate'):
        np.testing.assert_array_equal(out[mode][0], out['loop'][0])
        np.testing.assert_array_equal(out[mode][1], out['loop'][1])
        assert out[mode][2] == out['loop'][2]


def test_default_arithmetic_and_the_fallback_warning(env):
    """likelihood_arithmetic='auto' (the default): bf16x3 where a kernel
    exists and one evaluation is >= 1e10 flop, the fp32 kernels for small
    (latency-bound) problems and for widths without a bf16x3 kernel --
    silently.  An explicit 'bf16x3' that cannot be honoured runs fp32 too,
    with a LikelihoodArithmeticWarning; `hmc.arithmetic_reason` says why."""
    import warnings
    torch, _capi, dev = env
    import zhusuan_amd as zs
    from zhusuan_amd import _ops
    g = torch.Generator(device=dev).manual_seed(0)

    def sampler(N, D, C, **kw):
        X = torch.randn(N, D, device=dev, generator=g)
        y = (torch.rand(N, device=dev, generator=g) < 0.5).float()

        @zs.meta_bayesian_net()
        def blr():
            bn = zs.BayesianNet()
            wn = bn.normal('w', torch.zeros(D, device=dev), std=1.,
                           n_samples=C, group_ndims=1)
            bn.bernoulli('y', wn.tensor @ X.t(), group_ndims=1,
                         dtype=torch.float32)
            return bn
        hmc = zs.HMC(step_size=1e-3, n_leapfrogs=2, seed=1, **kw)
        w = torch.zeros(C, D, device=dev)
        op, info = hmc.sample(blr(), {'y': y}, {'w': w})
        assert hmc.plan_kind == 'linear_bernoulli'
        op.run()
        assert bool(torch.isfinite(info.log_prob).all())
        return hmc

    with warnings.catch_warnings():
        warnings.simplefilter('error', zs.LikelihoodArithmeticWarning)
        # 4 * 20 000 * 128 * 2 048 = 2.1e10 flop: the matrix cores' problem
        big = sampler(20000, 100, 2048)
        assert big.likelihood_arithmetic == 'auto'
        assert big.likelihood_arithmetic_used == 'bf16x3'
        assert big.arithmetic_reason is None
        # 4 * 500 * 64 * 256 = 3.3e7 flop: latency-bound
        small = sampler(500, 40, 256)
        assert small.likelihood_arithmetic_used == 'fp32'
        assert 'latency-bound' in small.arithmetic_reason
        # asked for by name, it is taken at any size
        assert sampler(500, 40, 256, likelihood_arithmetic='bf16x3'
                       ).likelihood_arithmetic_used == 'bf16x3'
        # no bf16x3 kernel past 256 padded columns: 'auto' says nothing
        wide = sampler(40000, 300, 1024)
        assert wide.likelihood_arithmetic_used == 'fp32'
        assert '320' in wide.arithmetic_reason
        assert sampler(500, 40, 256, likelihood_arithmetic='fp32'
                       ).arithmetic_reason is None
    with pytest.warns(zs.LikelihoodArithmeticWarning, match='<= 256 padded'):
        wide = sampler(2000, 300, 128, likelihood_arithmetic='bf16x3')
    assert wide.likelihood_arithmetic_used == 'fp32'
    with pytest.raises(ValueError, match="'auto', 'fp32' or 'bf16x3'"):
        zs.HMC(likelihood_arithmetic='fp16')
    assert _ops.BF16X3_AUTO_MIN_FLOP == 1.0e10


def test_hmc_topic_model_runs_over_the_documents_own_vocabularies(env,
                                                                  monkeypatch):
    """A topic model whose chain axis fills one-document workgroups, on the
    bf16x3 kernel: the plan hands the kernel the documents' OWN vocabularies
    (sparse bag-of-words counts; ABI 0.6.0) -- through sample_op.run (Python
    launch loop) and run_many (zshmc_hmc_model_run: the descriptor's obs_sp_*
    fields) bit for bit, and within float32 summation noise of the dense form
    of the same kernel and of the fp32 kernel."""
    torch, _capi, dev = env
    import zhusuan_amd as zs
    rng = np.random.RandomState(8)
    n_chains, n_docs, K, V = 128, 3, 20, 900
    beta = torch.tensor(rng.normal(size=(K, V)).astype(np.float32), device=dev)
    x = rng.poisson(0.06, size=(n_docs, V)).astype(np.float32)
    x[2] = 0.0
    x_t = torch.tensor(x, device=dev)
    eta0 = (0.3 * rng.normal(size=(n_chains, n_docs, K))).astype(np.float32)
    phi = torch.softmax(beta, -1)

    out = {}
    for mode, arithmetic, fill in (('loop', 'bf16x3', 0.6),
                                   ('block', 'bf16x3', 0.6),
                                   ('loop', 'bf16x3', 0.0),
                                   ('loop', 'fp32', 0.6)):
        monkeypatch.setattr(zs._ops, 'BF16X3_SPARSE_MAX_FILL', fill)
        zs._ops.clear_caches()

        @zs.meta_bayesian_net(scope='lntm')
        def lntm():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', torch.zeros(n_docs, K, device=dev),
                            logstd=0., n_samples=n_chains, group_ndims=1)
            bn.unnormalized_multinomial(
                'x', zs.log_mixture(torch.softmax(eta.tensor, -1), phi),
                normalize_logits=False, dtype=torch.float32)
            return bn
        m = lntm()
        m.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                  bn.cond_log_prob('x'))
        hmc = zs.HMC(step_size=0.02, n_leapfrogs=4, seed=3,
                     likelihood_arithmetic=arithmetic)
        eta = torch.tensor(eta0, device=dev)
        op, info = hmc.sample(m, {'x': x_t}, {'eta': eta})
        assert hmc.plan_kind == 'mixture_multinomial'
        assert hmc.likelihood_arithmetic_used == arithmetic
        own = hmc._plan.obs_sp is not None and not hmc._plan.sparse_rows
        assert own == (arithmetic == 'bf16x3' and fill > 0), (mode, fill)
        # (exact fp32, few rows, sparse counts: the row-by-row vector-ALU form,
        # tests/test_gpu_sparse_multinomial.py)
        assert hmc._plan.sparse_rows == (arithmetic == 'fp32')
        if own:
            assert hmc._plan.n_inner_run < V // 4
        if mode == 'block':
            op.run_many(3)
        else:
            for _ in range(3):
                op.run()
        out[(mode, arithmetic, fill)] = (eta.cpu().numpy(),
                                         info.log_prob.cpu().numpy())
    a = out[('loop', 'bf16x3', 0.6)]
    b = out[('block', 'bf16x3', 0.6)]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for other in (out[('loop', 'bf16x3', 0.0)], out[('loop', 'fp32', 0.6)]):
        close = np.isclose(a[0], other[0], atol=2e-4).reshape(
            n_chains * n_docs, -1).all(1)
        assert close.mean() > 0.98       # a borderline accept may flip
        np.testing.assert_allclose(a[1][close.reshape(a[1].shape)],
                                   other[1][close.reshape(a[1].shape)],
                                   rtol=2e-5, atol=2e-3)


def test_partly_filled_workgroups_over_own_vocabularies(env):
    """16 chains x 600 documents with sparse counts: neither the chain axis
    fills one-document workgroups nor is the problem small -- a workgroup of
    16 valid chains over the document's ~60 words beats 128 packed rows over
    all 3 000 (fill * 128 / 16 < 0.6): the own-vocabulary form, against the
    packed-rows form asked for through the fill bound, within summation
    noise."""
    torch, _capi, dev = env
    import zhusuan_amd as zs
    g = torch.Generator(device=dev).manual_seed(1)
    n_chains, n_docs, K, V = 16, 1200, 64, 3000
    phi = torch.softmax(torch.randn(K, V, device=dev, generator=g), -1)
    x = torch.poisson(torch.full((n_docs, V), 0.02, device=dev), generator=g)
    mean = torch.zeros(n_docs, K, device=dev)
    eta0 = 0.2 * torch.randn(n_chains, n_docs, K, device=dev, generator=g)
    out = {}
    for fill in (0.6, 0.0):
        zs._ops.clear_caches()
        old = zs._ops.BF16X3_SPARSE_MAX_FILL
        zs._ops.BF16X3_SPARSE_MAX_FILL = fill
        try:
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
            hmc = zs.HMC(step_size=0.02, n_leapfrogs=3, seed=2,
                         likelihood_arithmetic='bf16x3')
            eta = eta0.clone()
            op, info = hmc.sample(m, {'x': x}, {'eta': eta})
            assert hmc.likelihood_arithmetic_used == 'bf16x3'
            plan = hmc._plan
            assert (plan.obs_sp is not None) == (fill > 0)
            assert plan.packed_rows == (fill == 0)
            op.run_many(2)
            out[fill] = (eta.cpu().numpy(), info.log_prob.cpu().numpy())
        finally:
            zs._ops.BF16X3_SPARSE_MAX_FILL = old
    a, b = out[0.6], out[0.0]
    close = np.isclose(a[0], b[0], atol=2e-4).reshape(n_chains * n_docs,
                                                      -1).all(1)
    assert close.mean() > 0.98
