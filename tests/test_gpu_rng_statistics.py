"""The device's own random draws -- zshmc_momentum (Philox4x32-7 + Box-Muller
on v_log / v_sqrt / v_sin / v_cos) and the MH uniforms inside zshmc_mh_accept
-- beyond first and second moments: equiprobable-bin chi^2 against the normal
CDF, moments to the fourth, lag-1 correlation (values and squares) along the
latent, chain and iteration axes, Bernoulli frequencies and serial
independence of the accept bits.  The word-level battery on the same
counters runs on the CPU (tests/test_oracle_philox_statistics.py)."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from zhusuan_amd import _capi
    assert torch.cuda.is_available()
    return torch, _capi, torch.device('cuda', 0)


def _bound(n):
    return 5.0 / np.sqrt(n)


def test_device_momentum_stream(env):
    torch, capi, dev = env
    C, D, T = 4096, 256, 8
    stream = capi.current_stream()
    z = torch.empty(T, C, D, device=dev)
    for t in range(T):
        capi.call('zshmc_momentum', z[t].data_ptr(), None, C, D, 0, 3, t + 1,
                  0, None, stream)
    z = z.double()
    n = z.numel()
    edges = torch.tensor(stats.norm.ppf(np.linspace(0, 1, 65)[1:-1]),
                         device=dev)
    counts = torch.bincount(torch.bucketize(z.reshape(-1), edges),
                            minlength=64).double().cpu().numpy()
    chi2 = ((counts - n / 64.0) ** 2 / (n / 64.0)).sum()
    assert stats.chi2.sf(chi2, 63) > 1e-5, chi2
    m = [float((z ** k).mean()) for k in (1, 2, 3, 4)]
    assert abs(m[0]) < 5 / np.sqrt(n)
    assert abs(m[1] - 1) < 5 * np.sqrt(2.0 / n)
    assert abs(m[2]) < 5 * np.sqrt(15.0 / n)
    assert abs(m[3] - 3) < 5 * np.sqrt(96.0 / n)
    assert float(z.abs().max()) < 6.7          # the tail ends at 6.66 sigma
    for a, b in ((z[:-1], z[1:]), (z[:, :-1], z[:, 1:]),
                 (z[:, :, :-1], z[:, :, 1:]), (z[:, :, 0::2], z[:, :, 1::2])):
        assert abs(float((a * b).mean())) < _bound(a.numel())
        r2 = float(((a ** 2 - 1) * (b ** 2 - 1)).mean()) / 2.0
        assert abs(r2) < _bound(a.numel()), r2


@pytest.mark.parametrize('a', [0.03125, 0.5, 0.8, 0.96875])
def test_device_mh_uniforms_through_the_accept_bits(env, a):
    """acc = exp(min(H0 - H1, 0)) = a for every chain (kinetic energies
    chosen so); accept = (u < a): the bits are Bernoulli(a), independent
    along the chain and the iteration axis."""
    torch, capi, dev = env
    C, T = 65536, 16
    stream = capi.current_stream()
    zero = torch.zeros(C, device=dev)
    kin_new = torch.full((C,), float(-np.log(a)), device=dev)
    bits = torch.empty(T, C, dtype=torch.uint8, device=dev)
    acc = torch.empty(C, device=dev)
    scratch = [torch.empty(C, device=dev) for _ in range(3)]
    acc_sum = torch.zeros(2, dtype=torch.float64, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    for t in range(T):
        capi.call('zshmc_mh_accept', zero.data_ptr(), zero.data_ptr(),
                  zero.data_ptr(), kin_new.data_ptr(), C, 0, 5, t + 1,
                  acc.data_ptr(), scratch[0].data_ptr(),
                  scratch[1].data_ptr(), scratch[2].data_ptr(),
                  bits[t].data_ptr(), acc_sum.data_ptr(), flags.data_ptr(),
                  stream)
    np.testing.assert_allclose(acc.cpu().numpy(), a, rtol=1e-6)
    b = bits.double()
    n = b.numel()
    sd = np.sqrt(a * (1 - a))
    assert abs(float(b.mean()) - a) < 5 * sd / np.sqrt(n)
    v = (b - a) / sd
    for x, y in ((v[:-1], v[1:]), (v[:, :-1], v[:, 1:])):
        assert abs(float((x * y).mean())) < _bound(x.numel())
    # runs of accepted chains: the count of 11 pairs along the chain axis
    pairs = float((b[:, :-1] * b[:, 1:]).mean())
    assert abs(pairs - a * a) < 5 * np.sqrt(a * a * (1 - a * a) /
                                            b[:, 1:].numel()) + 1e-12
