"""The model layer as the reference's own tests exercise it
(tests/framework/test_base.py of the reference, the parts that belong to the
current -- non-legacy -- API), with sampling and log-densities on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import zhusuan_amd as zs
    assert torch.cuda.is_available()
    return zs, torch, torch.device('cuda', 0)


def test_query_surface(env):
    """test_base.py:130-209 (`get`, `cond_log_prob`, by name, list and
    iterator; a model function may return extra values next to the net)."""
    zs, torch, dev = env

    @zs.meta_bayesian_net()
    def build_meta_bn():
        bn = zs.BayesianNet()
        a = bn.normal('a', 0., logstd=1.)
        b = bn.normal('b', 0., logstd=1.)
        c = bn.normal('c', b, logstd=1.)
        return bn, a, b, c

    a_observed = torch.zeros([], device=dev)
    model, a, b, c = build_meta_bn().observe(a=a_observed)
    assert model.get('b') is b
    assert model.get(['b', 'c']) == [b, c]
    assert model.get(iter(['b', 'c'])) == [b, c]
    assert model['c'] is c
    assert a.is_observed() and a.tensor is a_observed
    assert not b.is_observed()
    # conditional log-densities: the node's own distribution at its value
    log_pa = model.cond_log_prob('a')
    np.testing.assert_allclose(float(log_pa), float(a.dist.log_prob(a_observed)),
                               atol=1e-6)
    np.testing.assert_allclose(float(log_pa), -0.5 * np.log(2 * np.pi) - 1.0,
                               atol=1e-6)
    log_pb, log_pc = model.cond_log_prob(['b', 'c'])
    np.testing.assert_allclose(float(log_pb), float(b.dist.log_prob(b.tensor)),
                               atol=1e-6)
    np.testing.assert_allclose(float(log_pc), float(c.dist.log_prob(c.tensor)),
                               atol=1e-6)
    log_pb2, log_pc2 = model.cond_log_prob(iter(['b', 'c']))
    assert float(log_pb2) == float(log_pb) and float(log_pc2) == float(log_pc)
    # c's mean IS b's sample (the node converts where a tensor is expected)
    np.testing.assert_allclose(float(c.dist.mean), float(b.tensor))
    # default log-joint = the sum over all stochastic nodes (bn.py:454-465)
    np.testing.assert_allclose(float(model.log_joint()),
                               float(log_pa + log_pb + log_pc), atol=1e-6)


def test_node_behaves_like_its_tensor(env):
    """test_base.py:69-92,94-108: arithmetic on a node, fetching a node."""
    zs, torch, dev = env

    @zs.meta_bayesian_net()
    def build():
        bn = zs.BayesianNet()
        bn.normal('a', 0., logstd=1.)
        bn.normal('t', torch.zeros(3, device=dev), std=1., n_samples=1)
        return bn
    one = torch.ones([], device=dev)
    samples = torch.tensor([1., 2., 3.], device=dev)
    bn = build().observe(a=one, t=samples)
    a, t = bn['a'], bn['t']
    assert float(a + 1) == 2.0 and float(1 + a) == 2.0
    assert float(a * 3 - 1) == 2.0 and float(-a) == -1.0
    assert float(torch.add(one, a.tensor)) == 2.0
    np.testing.assert_array_equal(zs.Session().run(t), [1, 2, 3])
    np.testing.assert_array_equal(t[1:].cpu().numpy(), [2, 3])


def test_duplicate_and_unknown_names(env):
    """test_base.py:124-128 and the lookup errors of bn.py:386-403."""
    zs, torch, dev = env
    bn = zs.BayesianNet()
    bn.normal('a', 0., logstd=1.)
    with pytest.raises(ValueError, match='Names should be unique'):
        bn.normal('a', 0., logstd=1.)
    with pytest.raises(ValueError, match="There isn't a node named 'zz'"):
        bn.get('zz')
    bn.deterministic('d', torch.zeros(2, device=dev))
    with pytest.raises(ValueError, match="Node 'd' is deterministic"):
        bn.cond_log_prob('d')
    with pytest.raises(TypeError, match='Expected string'):
        bn.get([1])


def test_observation_checks(env):
    """test_base.py:57-67: dtype and shape of an observation against the
    node's distribution (messages of bn.py:94-115)."""
    zs, torch, dev = env

    @zs.meta_bayesian_net()
    def build():
        bn = zs.BayesianNet()
        bn.normal('a', torch.zeros(2, device=dev), logstd=1.)
        return bn
    with pytest.raises(ValueError,
                       match=r"Incompatible shapes of StochasticTensor\('a'\)"):
        build().observe(a=torch.zeros(3, device=dev))
    # a broadcastable observation is fine (issue-49 shape: [1, 2] parameters)

    @zs.meta_bayesian_net()
    def issue_49():
        bn = zs.BayesianNet()
        bn.normal('x', torch.zeros(1, 2, device=dev),
                  logstd=torch.zeros(1, 2, device=dev), group_ndims=1)
        return bn
    x = issue_49().observe()['x']
    assert tuple(x.tensor.shape) == (1, 2)
    assert tuple(issue_49().observe(x=torch.zeros(1, 2, device=dev))
                 .cond_log_prob('x').shape) == (1,)


def test_meta_bn_reuse_and_scope_arguments(env):
    """test_base.py:232-268: scope / reuse_variables are accepted (there are
    no TensorFlow variables to scope here); the misuse the reference rejects
    is rejected with its message (meta_bn.py:131-134)."""
    zs, torch, dev = env

    @zs.meta_bayesian_net(scope='scp', reuse_variables=True)
    def build():
        bn = zs.BayesianNet()
        bn.normal('a', 0., logstd=1.)
        return bn
    m = build()
    assert m.observe()['a'].name == 'a'
    with pytest.raises(ValueError, match='Cannot reuse'):
        @zs.meta_bayesian_net(reuse_variables=True)
        def bad():
            return zs.BayesianNet()
        bad()
