"""GPU parity: stand-alone log_prob / gradient / sampling kernels
(csrc/distributions.hip) through the Python mirror of zhusuan.distributions
vs (1) the reference's own test vectors (tests/golden/logprob_vectors.json)
and (2) the NumPy oracle on seeded random inputs, including broadcast and
group_ndims cases.  Tolerance: the reference's assertAllClose default
rtol = atol = 1e-6 on the golden vectors (2e-6 where float32 exp/log of the
device differ in the last ulp); 1e-5 relative on random inputs."""
import json
import os

import numpy as np
import pytest

from oracle import distributions_ref as R
from oracle import philox

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


@pytest.fixture(scope='module')
def vectors(golden_dir):
    with open(os.path.join(golden_dir, 'logprob_vectors.json')) as f:
        return json.load(f)


def T(torch, dev, a, dtype=None):
    return torch.tensor(np.asarray(a), device=dev, dtype=dtype)


def test_normal_golden(env, vectors):
    zs, torch, dev = env
    for v in vectors['normal']:
        mean = T(torch, dev, v['mean'], torch.float32)
        logstd = T(torch, dev, v['logstd'], torch.float32)
        given = T(torch, dev, v['given'], torch.float32)
        tgt = np.array(v['log_prob'])
        d1 = zs.distributions.Normal(mean, logstd=logstd)
        np.testing.assert_allclose(d1.log_prob(given).cpu().numpy(), tgt,
                                   rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(d1.prob(given).cpu().numpy(), np.exp(tgt),
                                   rtol=1e-5, atol=1e-6)
        d2 = zs.distributions.Normal(mean, std=torch.exp(logstd))
        np.testing.assert_allclose(d2.log_prob(given).cpu().numpy(), tgt,
                                   rtol=2e-6, atol=2e-6)


def test_bernoulli_golden(env, vectors):
    zs, torch, dev = env
    for v in vectors['bernoulli']:
        d = zs.distributions.Bernoulli(T(torch, dev, v['logits'],
                                         torch.float32))
        got = d.log_prob(T(torch, dev, v['given'], torch.int32))
        np.testing.assert_allclose(got.cpu().numpy(), np.array(v['log_prob']),
                                   rtol=2e-6, atol=2e-6)


def test_categorical_golden(env, vectors):
    zs, torch, dev = env
    for v in vectors['categorical']:
        d = zs.distributions.Categorical(T(torch, dev, v['logits'],
                                           torch.float32))
        got = d.log_prob(T(torch, dev, v['given'], torch.int32))
        np.testing.assert_allclose(got.cpu().numpy(), np.array(v['log_prob']),
                                   rtol=2e-6, atol=2e-6)


def test_unnormalized_multinomial_golden(env, vectors):
    zs, torch, dev = env
    for v in vectors['unnormalized_multinomial']:
        d = zs.distributions.UnnormalizedMultinomial(
            T(torch, dev, v['logits'], torch.float32),
            normalize_logits=v['normalize'])
        got = d.log_prob(T(torch, dev, v['given'], torch.int32))
        tgt = np.array(v['log_prob'])
        np.testing.assert_allclose(got.cpu().numpy(), tgt, rtol=3e-6,
                                   atol=1e-2 if np.abs(tgt).max() > 1e4
                                   else 2e-6)


CASES = [  # (x shape, mean shape, logstd shape, group_ndims)
    ((7, 5), (5,), (5,), 0), ((7, 5), (5,), (5,), 1), ((7, 5), (), (), 1),
    ((6, 4, 3), (4, 3), (3,), 2), ((6, 4, 3), (6, 1, 3), (4, 1), 1),
    ((300, 257), (257,), (300, 257), 1), ((5,), (5,), (5,), 0),
    ((2, 3, 4, 5), (5,), (1,), 3), ((1000, 64), (1, 64), (64,), 1),
]


@pytest.mark.parametrize('xs,ms,ss,g', CASES)
def test_normal_random_forward_backward(env, xs, ms, ss, g):
    zs, torch, dev = env
    rng = np.random.RandomState(len(xs) * 7 + g)
    x = rng.normal(size=xs).astype(np.float32)
    mean = rng.normal(size=ms).astype(np.float32)
    logstd = (rng.normal(size=ss) * 0.5).astype(np.float32)
    ref = R.Normal(mean, logstd=logstd, group_ndims=g)
    xt = T(torch, dev, x).requires_grad_(True)
    mt = T(torch, dev, mean).requires_grad_(True)
    st = T(torch, dev, logstd).requires_grad_(True)
    d = zs.distributions.Normal(mt, logstd=st, group_ndims=g)
    lp = d.log_prob(xt)
    want = ref.log_prob(x)
    assert tuple(lp.shape) == want.shape
    np.testing.assert_allclose(lp.detach().cpu().numpy(), want, rtol=2e-5,
                               atol=2e-5 * max(1.0, np.abs(want).max()))
    w = rng.normal(size=want.shape).astype(np.float32)
    (lp * T(torch, dev, w)).sum().backward()
    full = np.broadcast(x, mean, logstd).shape
    wfull = np.broadcast_to(
        w.reshape(w.shape + (1,) * g) if g else w, full)
    gx = ref.grad_given(x) * wfull
    gm, gs = ref.grad_params(x)

    def reduce_to(a, shape):
        a = np.asarray(a * wfull, np.float64)
        while a.ndim > len(shape):
            a = a.sum(0)
        for ax, s in enumerate(shape):
            if s == 1 and a.shape[ax] != 1:
                a = a.sum(ax, keepdims=True)
        return a
    np.testing.assert_allclose(xt.grad.cpu().numpy(), reduce_to(
        ref.grad_given(x), xs), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(mt.grad.cpu().numpy(), reduce_to(gm, ms),
                               rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(st.grad.cpu().numpy(), reduce_to(gs, ss),
                               rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('ls,zs_,g', [((8, 6), (8, 6), 0), ((6,), (8, 6), 1),
                                      ((4, 1, 6), (5, 6), 2),
                                      ((), (7,), 1), ((500, 300), (500, 300), 1)])
def test_bernoulli_random_forward_backward(env, ls, zs_, g):
    zs, torch, dev = env
    rng = np.random.RandomState(sum(ls) + g)
    logits = (rng.normal(size=ls) * 3).astype(np.float32)
    given = (rng.uniform(size=zs_) < 0.5).astype(np.int32)
    ref = R.Bernoulli(logits, group_ndims=g)
    lt = T(torch, dev, logits).requires_grad_(True)
    d = zs.distributions.Bernoulli(lt, group_ndims=g)
    lp = d.log_prob(T(torch, dev, given))
    want = ref.log_prob(given)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), want, rtol=2e-5,
                               atol=2e-5 * max(1.0, np.abs(want).max()))
    lp.sum().backward()
    gl = np.asarray(ref.grad_logits(given), np.float64)
    while gl.ndim > len(ls):
        gl = gl.sum(0)
    for ax, s in enumerate(ls):
        if s == 1 and gl.shape[ax] != 1:
            gl = gl.sum(ax, keepdims=True)
    np.testing.assert_allclose(lt.grad.cpu().numpy(), gl, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('shape,ncat,g', [((9,), 3, 0), ((4, 5), 70, 1),
                                          ((3, 1), 129, 0), ((200,), 1000, 0)])
def test_categorical_random_forward_backward(env, shape, ncat, g):
    zs, torch, dev = env
    rng = np.random.RandomState(ncat)
    logits = (rng.normal(size=shape + (ncat,)) * 2).astype(np.float32)
    labels = rng.randint(0, ncat, size=shape)
    ref = R.Categorical(logits, group_ndims=g)
    lt = T(torch, dev, logits).requires_grad_(True)
    d = zs.distributions.Categorical(lt, group_ndims=g)
    lp = d.log_prob(T(torch, dev, labels))
    np.testing.assert_allclose(lp.detach().cpu().numpy(),
                               ref.log_prob(labels), rtol=2e-5, atol=2e-5)
    lp.sum().backward()
    np.testing.assert_allclose(lt.grad.cpu().numpy(), ref.grad_logits(labels),
                               rtol=2e-4, atol=2e-5)
    # labels broadcast against logits (univariate.py:499-505)
    lab2 = rng.randint(0, ncat, size=(2,) + shape)
    np.testing.assert_allclose(
        zs.distributions.Categorical(lt.detach()).log_prob(
            T(torch, dev, lab2)).cpu().numpy(),
        R.Categorical(logits).log_prob(lab2), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('normalize', [True, False])
def test_unnormalized_multinomial_random(env, normalize):
    zs, torch, dev = env
    rng = np.random.RandomState(5)
    logits = rng.normal(size=(6, 11, 300)).astype(np.float32)
    cnt = rng.poisson(2.0, size=(6, 11, 300)).astype(np.int32)
    ref = R.UnnormalizedMultinomial(logits, normalize_logits=normalize,
                                    group_ndims=1)
    lt = T(torch, dev, logits).requires_grad_(True)
    d = zs.distributions.UnnormalizedMultinomial(
        lt, normalize_logits=normalize, group_ndims=1)
    lp = d.log_prob(T(torch, dev, cnt))
    want = ref.log_prob(cnt)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), want, rtol=3e-5,
                               atol=3e-5 * np.abs(want).max())
    lp.sum().backward()
    np.testing.assert_allclose(lt.grad.cpu().numpy(), ref.grad_logits(cnt),
                               rtol=3e-4, atol=3e-4)


def test_empty_inputs(env):
    zs, torch, dev = env
    d = zs.distributions.Normal(torch.zeros(0, 4, device=dev),
                                std=torch.ones(4, device=dev), group_ndims=1)
    assert tuple(d.log_prob(torch.zeros(0, 4, device=dev)).shape) == (0,)
    b = zs.distributions.Bernoulli(torch.zeros(0, device=dev))
    assert tuple(b.log_prob(torch.zeros(0, device=dev)).shape) == (0,)


def test_device_philox_matches_oracle(env):
    """Momentum kernel = Philox4x32-7 + Box-Muller on the hardware
    log/sqrt/sin/cos units; uniform bits are exact, normals agree with the
    float64-evaluated oracle to a few 1e-6."""
    zs, torch, dev = env
    from zhusuan_amd import _capi
    for C, D, off, it, lat in [(33, 10, 0, 1, 0), (7, 1030, 5000, 77, 2),
                               (64, 4, 2 ** 31, 2 ** 31 + 5, 1)]:
        p = torch.empty(C, D, device=dev)
        kin = torch.zeros(C, device=dev)
        _capi.call('zshmc_momentum', p.data_ptr(), None, C, D, off, 0xABCDEF0123,
                   it & 0xFFFFFFFF, lat, kin.data_ptr(), _capi.current_stream())
        want = philox.normal_chain_major(0xABCDEF0123, it, C, D,
                                         chain_offset=off, latent_id=lat)
        got = p.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=4e-6)
        np.testing.assert_allclose(kin.cpu().numpy(),
                                   0.5 * (want.astype(np.float64) ** 2).sum(1),
                                   rtol=1e-4)


def test_sampling_matches_oracle_stream(env):
    zs, torch, dev = env
    zs.set_random_seed(1234)
    mean = np.linspace(-1, 1, 6).astype(np.float32).reshape(2, 3)
    std = np.array([0.5, 1.0, 2.0], np.float32)
    n = zs.distributions.Normal(T(torch, dev, mean), std=T(torch, dev, std))
    s0 = n.sample(50)                  # op offset 0
    s1 = n.sample()                    # op offset 1, squeezed
    assert tuple(s0.shape) == (50, 2, 3) and tuple(s1.shape) == (2, 3)
    r = R.Normal(mean, std=std)
    np.testing.assert_allclose(s0.cpu().numpy(), r.sample(50, seed=1234,
                                                          offset=0),
                               rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(s1.cpu().numpy(), r.sample(None, seed=1234,
                                                          offset=1),
                               rtol=2e-5, atol=1e-5)
    logits = np.array([[-2.0, 0.0, 3.0], [1.0, -1.0, 0.5]], np.float32)
    b = zs.distributions.Bernoulli(T(torch, dev, logits))
    sb = b.sample(4000)                # offset 2
    rb = R.Bernoulli(logits).sample(4000, seed=1234, offset=2)
    assert sb.dtype == torch.int32 and tuple(sb.shape) == (4000, 2, 3)
    assert (sb.cpu().numpy() != rb).mean() < 1e-3
    c = zs.distributions.Categorical(T(torch, dev, logits))
    sc = c.sample(4000)                # offset 3
    rc = R.Categorical(logits).sample(4000, seed=1234, offset=3)
    assert tuple(sc.shape) == (4000, 2)
    assert (sc.cpu().numpy() != rc).mean() < 1e-3
    freq = np.bincount(sc.cpu().numpy()[:, 0], minlength=3) / 4000.0
    soft = np.exp(logits[0]) / np.exp(logits[0]).sum()
    np.testing.assert_allclose(freq, soft, atol=0.03)
