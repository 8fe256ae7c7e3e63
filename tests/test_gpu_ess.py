"""GPU parity of the batched ESS kernels (csrc/diagnostics.hip) against the
oracle restatement of the reference's zhusuan/diagnostics.py and against the
fixtures produced by the reference's own module (tests/golden/ess_fixture.npz)."""
import os

import numpy as np
import pytest

from oracle import ess_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _ar1(rng, n, s, phi):
    x = np.zeros((n, s), np.float32)
    e = rng.normal(size=(n, s)).astype(np.float32)
    for i in range(1, n):
        x[i] = phi * x[i - 1] + e[i]
    return x


# n <= 512: LDS-staged path; n > 512: global re-read path
@pytest.mark.parametrize('n,chains,dims,phi', [(300, 5, 7, 0.3), (64, 3, 130, 0.0),
                                             (512, 2, 64, 0.8), (900, 4, 3, 0.6),
                                             (2, 1, 5, 0.0)])
def test_series_match_oracle(env, n, chains, dims, phi):
    zs, torch, dev = env
    rng = np.random.RandomState(n + dims)
    x = _ar1(rng, n, chains * dims, np.linspace(-0.5, phi, chains * dims)
             .astype(np.float32)).reshape(n, chains, dims)
    xt = torch.tensor(x, device=dev)
    per = zs.diagnostics.effective_sample_size_device(xt, burn_in=0,
                                                       per_dimension=True)
    ref = np.array([[ess_ref.effective_sample_size_1d(x[:, c, d].astype(np.float64))
                     for d in range(dims)] for c in range(chains)])
    np.testing.assert_allclose(per.cpu().numpy(), ref, rtol=2e-5)
    # min positive over dims, as diagnostics.py:55-64, with burn-in
    burn = min(10, n - 2)
    got = zs.diagnostics.effective_sample_size_device(xt, burn_in=burn)
    want = [ess_ref.effective_sample_size(x[:, c, :].astype(np.float64), burn_in=burn)
            for c in range(chains)]
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5)
    # the host FFT version agrees too
    np.testing.assert_allclose(
        per.cpu().numpy(),
        zs.diagnostics.effective_sample_size_batch(x, burn_in=0), rtol=2e-5)


def test_reference_fixtures(env):
    zs, torch, dev = env
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden',
                              'ess_fixture.npz'))
    for case in ('iid', 'ar1', 'rwmh', 'sticky_f32'):
        s = fx[case + '_samples'].astype(np.float32)
        xt = torch.tensor(s[:, None, :], device=dev)       # one chain
        per = zs.diagnostics.effective_sample_size_device(xt, burn_in=0,
                                                           per_dimension=True)
        np.testing.assert_allclose(per.cpu().numpy()[0], fx[case + '_ess1d'],
                                   rtol=5e-5)
        for burn in (0, 100):
            got = zs.diagnostics.effective_sample_size_device(xt, burn_in=burn)
            np.testing.assert_allclose(float(got[0]),
                                       fx['%s_ess_burn%d' % (case, burn)],
                                       rtol=5e-5)


def test_rejects_host_tensors_and_short_series(env):
    zs, torch, dev = env
    with pytest.raises(TypeError):
        zs.diagnostics.effective_sample_size_device(torch.zeros(10, 2, 2))
    with pytest.raises(Exception, match='at least 2 draws'):
        zs.diagnostics.effective_sample_size_device(
            torch.zeros(1, 2, 2, device=dev), burn_in=0)
