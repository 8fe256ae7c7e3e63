"""The C entry points behind the native plan's PACKED state, called straight
through the C-ABI (include/zshmc.h): zshmc_momentum_rows and zshmc_copy_rows
(a latent <-> its columns of rows that are `ld` floats apart) and
zshmc_model_kick_drift with n_data < row_stride -- the last 16-byte group of a
row partly padding, which must stay out of the prior (univariate.py:174-181
summed by group_ndims = 1) and of the softmax (lntm_mcem.py:39) and must stay
zero.  Checked against each other bit for bit, against the oracle's generator
(to the device transcendentals' last bit) and against a float64 NumPy
restatement of one leapfrog trip (hmc.py:38-43, :352-364)."""
import numpy as np
import pytest

from oracle import philox

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from zhusuan_amd import _capi
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    return torch, _capi, dev, torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize('D,ld,off', [(13, 20, 0), (13, 20, 5), (16, 16, 0),
                                      (256, 300, 8), (1, 4, 3), (7, 1004, 997)])
def test_momentum_rows_lays_down_the_contiguous_draw(env, D, ld, off):
    """Same Philox counters as zshmc_momentum (chain, group of 4 columns,
    iteration, latent): the strided form writes the SAME numbers into columns
    [off, off + D) of rows `ld` apart, touches nothing else, and adds the
    same kinetic energy."""
    torch, capi, dev, s = env
    C, seed, it, latent, chain_offset = 77, 1234567, 9, 2, 1000
    mass = torch.rand(D, device=dev) + 0.5
    for use_mass in (False, True):
        m = mass if use_mass else None
        p_ref = torch.empty(C, D, device=dev)
        kin_ref = torch.zeros(C, device=dev)
        capi.call('zshmc_momentum', p_ref.data_ptr(), capi.ptr(m), C, D,
                  chain_offset, seed, it, latent, kin_ref.data_ptr(), s)
        buf = torch.full((C, ld), 7.0, device=dev)
        kin = torch.ones(C, device=dev)          # accumulates: starts at 1
        capi.call('zshmc_momentum_rows', buf.data_ptr() + 4 * off, ld,
                  capi.ptr(m), C, D, chain_offset, seed, it, latent,
                  kin.data_ptr(), s)
        assert torch.equal(buf[:, off:off + D], p_ref)
        rest = torch.cat([buf[:, :off], buf[:, off + D:]], 1)
        assert bool((rest == 7.0).all())
        torch.testing.assert_close(kin - 1.0, kin_ref, rtol=1e-5, atol=1e-5)
        # and both are the oracle's draw (hmc.py:21-23)
        z = philox.normal_chain_major(seed, it, C, D, chain_offset=chain_offset,
                                      latent_id=latent)
        # (the device's log / sin / cos differ from NumPy's in the last bit)
        want = z * np.sqrt(mass.cpu().numpy()) if use_mass else z
        np.testing.assert_allclose(p_ref.cpu().numpy(), want, rtol=3e-6,
                                   atol=1e-6)
    with pytest.raises(capi.ZshmcError):
        capi.call('zshmc_momentum_rows', buf.data_ptr(), D - 1 if D > 1 else 0,
                  None, C, D, 0, seed, it, 0, None, s)


@pytest.mark.parametrize('n,ld_dst,ld_src', [(13, 13, 20), (13, 20, 13),
                                             (16, 16, 32), (300, 300, 1004),
                                             (1, 1, 4)])
def test_copy_rows_with_and_without_accept(env, n, ld_dst, ld_src):
    torch, capi, dev, s = env
    R = 131
    g = torch.Generator(device=dev).manual_seed(n)
    src = torch.randn(R, ld_src, device=dev, generator=g)
    dst0 = torch.randn(R, ld_dst, device=dev, generator=g)
    off_s, off_d = (ld_src - n) // 2, (ld_dst - n) // 2
    for accept in (None, (torch.rand(R, device=dev, generator=g) < 0.4)
                   .to(torch.uint8)):
        dst = dst0.clone()
        capi.call('zshmc_copy_rows', dst.data_ptr() + 4 * off_d, ld_dst,
                  src.data_ptr() + 4 * off_s, ld_src, capi.ptr(accept), R, n,
                  s)
        want = dst0.clone()
        rows = slice(None) if accept is None else accept.bool()
        want[rows, off_d:off_d + n] = src[rows, off_s:off_s + n]
        assert torch.equal(dst, want)
    with pytest.raises(capi.ZshmcError):
        capi.call('zshmc_copy_rows', dst.data_ptr(), n - 1 if n > 1 else 0,
                  src.data_ptr(), ld_src, None, R, n, s)


def _trip_ref(q, p, g_lik, theta, mean, logstd, mass, eps, kick, drift, scale,
              ll, softmax):
    """One launch of zshmc_model_kick_drift in float64 (valid columns only)."""
    q, p = q.astype(np.float64), p.astype(np.float64)
    prec = np.exp(-2.0 * logstd.astype(np.float64))
    r = q - mean
    prior = (-0.5 * np.log(2 * np.pi) - logstd - 0.5 * prec * r * r).sum(1)
    gl = scale * g_lik.astype(np.float64)
    if softmax:
        th = theta.astype(np.float64)
        gl = th * (gl - (gl * th).sum(1, keepdims=True))
    grad = gl - prec * r
    p = p + kick * eps * grad
    vel = p / mass
    if drift != 0:
        q = q + drift * eps * vel
    kin = 0.5 * (p * vel).sum(1)
    lp = scale * ll + prior
    if softmax:
        e = np.exp(q - q.max(1, keepdims=True))
        op = e / e.sum(1, keepdims=True)
    else:
        op = q
    return q, p, lp, kin, op


@pytest.mark.parametrize('softmax', [0, 1])
@pytest.mark.parametrize('D,ld,width,rows_m,rows_l', [
    (37, 40, 64, 1, 1), (6, 8, 64, 5, 1), (130, 132, 256, 1, 5),
    (301, 304, 512, 1, 1), (1001, 1004, 1024, 1, 1), (16, 16, 64, 1, 1),
    (1, 4, 64, 1, 1)])
def test_model_kick_drift_on_padded_rows(env, softmax, D, ld, width, rows_m,
                                         rows_l):
    torch, capi, dev, s = env
    C = 50
    rng = np.random.RandomState(D + softmax)
    f32 = np.float32

    def padded(a, cols):
        out = np.zeros((a.shape[0], cols), f32)
        out[:, :a.shape[1]] = a
        return out
    q = (0.5 * rng.normal(size=(C, D))).astype(f32)
    p = rng.normal(size=(C, D)).astype(f32)
    g_lik = rng.normal(size=(C, D)).astype(f32)
    mean = (0.3 * rng.normal(size=(rows_m, D))).astype(f32)
    logstd = (0.2 * rng.normal(size=(rows_l, D))).astype(f32)
    mass = (rng.uniform(0.5, 2.0, size=D)).astype(f32)
    ll = rng.normal(size=C).astype(f32)
    e = np.exp(q - q.max(1, keepdims=True))
    theta = (e / e.sum(1, keepdims=True)).astype(f32)
    eps, kick, drift, scale = 0.05, 0.5, 1.0, 0.7
    T = lambda a: torch.tensor(a, device=dev)
    q_t, p_t = T(padded(q, ld)), T(padded(p, ld))
    g_t = T(padded(g_lik, width))
    # the operand carries theta (softmax: read AND written) / receives q'
    op_t = T(padded(theta, width)) if softmax else torch.full(
        (C, width), 9.0, device=dev)
    m_t = T(padded(mean, ld))
    l_t = T(padded(logstd, ld))
    mass_p = np.ones(ld, f32)
    mass_p[:D] = mass
    mass_t = T(mass_p)
    ll_t = T(ll)
    lp_t = torch.zeros(C, device=dev)
    kin_t = torch.ones(C, device=dev)
    capi.call('zshmc_model_kick_drift', q_t.data_ptr(), p_t.data_ptr(),
              g_t.data_ptr(), width, op_t.data_ptr(), width, softmax,
              m_t.data_ptr(), rows_m, l_t.data_ptr(), rows_l,
              mass_t.data_ptr(), None, eps, kick, drift, scale, C, D, ld,
              ll_t.data_ptr(), lp_t.data_ptr(), kin_t.data_ptr(), s)
    rows = np.arange(C)
    q1, p1, lp, kin, op = _trip_ref(
        q, p, g_lik, theta, mean[rows % rows_m], logstd[rows % rows_l], mass,
        eps, kick, drift, scale, ll, softmax)
    got_q, got_p = q_t.cpu().numpy(), p_t.cpu().numpy()
    np.testing.assert_allclose(got_q[:, :D], q1, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(got_p[:, :D], p1, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(lp_t.cpu().numpy(), lp, rtol=3e-6,
                               atol=3e-6 * D)
    np.testing.assert_allclose(kin_t.cpu().numpy() - 1.0, kin, rtol=3e-6,
                               atol=3e-6 * D)
    got_op = op_t.cpu().numpy()
    np.testing.assert_allclose(got_op[:, :D], op, rtol=3e-6, atol=3e-6)
    # the padding: untouched zeros in q and p, zeros in the operand
    assert not got_q[:, D:].any() and not got_p[:, D:].any()
    assert not got_op[:, D:].any()
    if softmax:
        np.testing.assert_allclose(got_op.sum(1), 1.0, rtol=1e-5)
    # a row stride below n_data, or one that is not a multiple of 4
    for bad in (D - 1 if D > 1 else 0, ld + 1):
        with pytest.raises(capi.ZshmcError):
            capi.call('zshmc_model_kick_drift', q_t.data_ptr(), p_t.data_ptr(),
                      g_t.data_ptr(), width, op_t.data_ptr(), width, softmax,
                      m_t.data_ptr(), rows_m, l_t.data_ptr(), rows_l,
                      mass_t.data_ptr(), None, eps, kick, drift, scale, C, D,
                      bad, ll_t.data_ptr(), lp_t.data_ptr(), kin_t.data_ptr(),
                      s)
