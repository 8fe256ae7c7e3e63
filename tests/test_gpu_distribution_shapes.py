"""Shape behaviour of the distributions on the HMC path, with the tables of
the reference's own tests (tests/distributions/utils.py: the fully defined
rows of test_batch_shape_*, test_*_sample_shape_*, test_*_log_prob_shape_*;
tests/distributions/test_base.py: group_ndims).  Partially defined TensorFlow
shapes (None) have no counterpart on concrete torch tensors."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def _z(torch, dev, shape):
    return torch.zeros(shape, device=dev)


# utils.py:186-229, :232-273, :276-315 (two-parameter univariate: Normal)
BATCH_2P = [([2, 3], [], [2, 3]), ([2, 3], [3], [2, 3]),
            ([2, 1, 4], [2, 3, 4], [2, 3, 4]), ([2, 3, 5], [3, 1], [2, 3, 5]),
            ([1, 2, 3], [1, 3], [1, 2, 3])]
SAMPLE_2P = [([2, 3], [], None, [2, 3]), ([2, 3], [], 1, [1, 2, 3]),
             ([5], [5], 2, [2, 5]), ([2, 1, 4], [1, 2, 4], 3, [3, 2, 2, 4]),
             ([2, 3], [2, 1], 1, [1, 2, 3]), ([1, 3], [], 2, [2, 1, 3]),
             ([2, 1, 5], [3, 1], 3, [3, 2, 3, 5])]
LOGP_2P = [([2, 3], [], [2, 3], [2, 3]), ([5], [5], [2, 1], [2, 5]),
           ([2, 3], [2, 1], [1, 3], [2, 3]), ([1, 3], [], [2, 1, 3], [2, 1, 3]),
           ([1, 5], [3, 1], [1, 2, 1, 1], [1, 2, 3, 5])]


def test_normal_shapes(env):
    zs, torch, dev = env
    for p1, p2, want in BATCH_2P:
        d = zs.distributions.Normal(_z(torch, dev, p1), logstd=_z(torch, dev, p2))
        assert list(d.get_batch_shape()) == want == list(d.batch_shape)
        assert list(d.get_value_shape()) == []
    with pytest.raises((ValueError, RuntimeError)):
        zs.distributions.Normal(_z(torch, dev, [2, 3, 5]),
                                logstd=_z(torch, dev, [3, 2])).sample(1)
    for p1, p2, n, want in SAMPLE_2P:
        d = zs.distributions.Normal(_z(torch, dev, p1), std=1 + _z(torch, dev, p2))
        assert list(d.sample(n).shape) == want
    for p1, p2, g, want in LOGP_2P:
        d = zs.distributions.Normal(_z(torch, dev, p1), logstd=_z(torch, dev, p2))
        assert list(d.log_prob(_z(torch, dev, g)).shape) == want
        assert list(d.prob(_z(torch, dev, g)).shape) == want
    d = zs.distributions.Normal(_z(torch, dev, [2, 3, 5]), logstd=0.)
    with pytest.raises(ValueError, match='broadcast to match'):
        d.log_prob(_z(torch, dev, [1, 2, 1]))


# utils.py:318-357, :360-395, :398-438 (one-parameter univariate: Bernoulli)
def test_bernoulli_shapes(env):
    zs, torch, dev = env
    for shape in ([], [2], [2, 3], [2, 1, 4]):
        d = zs.distributions.Bernoulli(_z(torch, dev, shape))
        assert list(d.get_batch_shape()) == shape
        assert list(d.get_value_shape()) == []
    for shape, n, want in (([2, 3], None, [2, 3]), ([2, 3], 1, [1, 2, 3]),
                           ([5], 2, [2, 5]), ([1, 3], 2, [2, 1, 3]),
                           ([2, 1, 5], 3, [3, 2, 1, 5])):
        s = zs.distributions.Bernoulli(_z(torch, dev, shape)).sample(n)
        assert list(s.shape) == want and s.dtype == torch.int32
    for shape, g, want in (([2, 3], [2, 1], [2, 3]), ([5], [2, 1], [2, 5]),
                           ([2, 3], [1, 3], [2, 3]),
                           ([1, 3], [2, 2, 3], [2, 2, 3]),
                           ([1, 5], [1, 2, 3, 1], [1, 2, 3, 5])):
        d = zs.distributions.Bernoulli(_z(torch, dev, shape))
        given = torch.zeros(g, dtype=torch.int32, device=dev)
        assert list(d.log_prob(given).shape) == want
    with pytest.raises(ValueError, match='broadcast to match'):
        zs.distributions.Bernoulli(_z(torch, dev, [2, 3, 5])).log_prob(
            torch.zeros([1, 2, 1], dtype=torch.int32, device=dev))


# utils.py:318-357 with is_univariate=False, :441-477 (Categorical: the value
# is a class index, batch shape = logits.shape[:-1])
def test_categorical_shapes(env):
    zs, torch, dev = env
    with pytest.raises(ValueError):
        zs.distributions.Categorical(_z(torch, dev, []))
    for shape in ([2], [2, 3], [2, 1, 4]):
        d = zs.distributions.Categorical(_z(torch, dev, shape))
        assert list(d.get_batch_shape()) == shape[:-1]
        assert list(d.get_value_shape()) == []
        assert d.n_categories == shape[-1]
    for shape, n, want in (([2], None, []), ([2], 1, [1]), ([2, 3], None, [2]),
                           ([2, 3], 1, [1, 2]), ([5], 2, [2]),
                           ([1, 2, 4], 3, [3, 1, 2]), ([2, 1, 5], 3, [3, 2, 1])):
        s = zs.distributions.Categorical(_z(torch, dev, shape)).sample(n)
        assert list(s.shape) == want
    # test_univariate.py: given [..] against logits [.., n_cat]
    for shape, g, want in (([2, 3], [2], [2]), ([2, 5], [1], [2]),
                           ([1, 2, 4], [1], [1, 2]), ([3, 1, 5], [1, 4], [3, 4]),
                           ([1, 4], [2, 5], [2, 5])):
        d = zs.distributions.Categorical(_z(torch, dev, shape))
        given = torch.zeros(g, dtype=torch.int32, device=dev)
        assert list(d.log_prob(given).shape) == want
    with pytest.raises(ValueError, match='broadcast to match'):
        zs.distributions.Categorical(_z(torch, dev, [2, 3, 5])).log_prob(
            torch.zeros([1, 2], dtype=torch.int32, device=dev))


# utils.py:480-520 (value has one axis: UnnormalizedMultinomial)
def test_unnormalized_multinomial_shapes(env):
    zs, torch, dev = env
    for shape, g, want in (([2, 3], [2, 3], [2]), ([2, 5], [5], [2]),
                           ([1, 2, 4], [4], [1, 2]),
                           ([3, 1, 5], [1, 4, 5], [3, 4]),
                           ([1, 4], [2, 5, 4], [2, 5])):
        d = zs.distributions.UnnormalizedMultinomial(_z(torch, dev, shape))
        assert list(d.get_batch_shape()) == shape[:-1]
        assert list(d.get_value_shape()) == [shape[-1]]
        given = torch.ones(g, dtype=torch.int32, device=dev)
        assert list(d.log_prob(given).shape) == want
    with pytest.raises(ValueError, match='broadcast to match'):
        zs.distributions.UnnormalizedMultinomial(
            _z(torch, dev, [2, 3, 5])).log_prob(
                torch.ones([1, 2, 5], dtype=torch.int32, device=dev))
    with pytest.raises(NotImplementedError):
        zs.distributions.UnnormalizedMultinomial(_z(torch, dev, [2, 3])).sample()


def test_group_ndims(env):
    """tests/distributions/test_base.py: log_prob sums the last group_ndims
    batch axes; a negative or too large group_ndims is rejected."""
    zs, torch, dev = env
    x = torch.randn(2, 3, 4, device=dev)
    base = zs.distributions.Normal(_z(torch, dev, [3, 4]), logstd=0.)
    full = base.log_prob(x)
    for g in (0, 1, 2):
        d = zs.distributions.Normal(_z(torch, dev, [3, 4]), logstd=0.,
                                    group_ndims=g)
        want = full if g == 0 else full.sum(tuple(range(-g, 0)))
        got = d.log_prob(x)
        assert list(got.shape) == list(want.shape)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        zs.distributions.Normal(0., logstd=0., group_ndims=-1)
