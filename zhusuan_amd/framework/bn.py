"""BayesianNet / StochasticTensor: named stochastic + deterministic nodes and
the joint log-probability contract HMC differentiates.  Mirrors reference
zhusuan/framework/bn.py:26-316 (StochasticTensor), :319-478 (_BayesianNet),
:556-590/:628-682/:938-965 (factory methods on the HMC path).  Values are
torch device tensors; log-probs come from the HIP kernels via
zhusuan_amd.distributions."""
import torch

from ..utils import broadcast_shapes

from .. import distributions
from ..distributions.base import as_tensor
from .meta_bn import active_frame

__all__ = ['StochasticTensor', 'BayesianNet']


class StochasticTensor(object):
    """bn.py:26-316.  Behaves like its `tensor` in torch expressions."""

    def __init__(self, bn, name, dist, observation=None, **kwargs):
        self._bn = bn
        self._name = name
        self._dist = dist
        self._dtype = dist.dtype
        self._n_samples = kwargs.get("n_samples", None)
        if observation is not None:
            self._observation = self._check_observation(observation)
        else:
            self._observation = None

    def _check_observation(self, observation):
        """bn.py:94-115 (messages kept)."""
        type_msg = "Incompatible types of {}('{}') and its observation: {}"
        try:
            observation = as_tensor(observation, dtype=self._dtype,
                                    device=self._dist._device(),
                                    keep_symbolic=True)
        except (ValueError, TypeError, RuntimeError) as e:
            raise ValueError(type_msg.format(self.__class__.__name__,
                                             self._name, e))
        shape_msg = "Incompatible shapes of {}('{}') and its observation: " \
                    "{} vs {}."
        dist_shape = tuple(self._dist.get_batch_shape()) + tuple(
            self._dist.get_value_shape())
        try:
            broadcast_shapes(dist_shape, tuple(observation.shape))
        except RuntimeError:
            raise ValueError(shape_msg.format(
                self.__class__.__name__, self._name, dist_shape,
                tuple(observation.shape)))
        return observation

    @property
    def bn(self):
        return self._bn

    @property
    def name(self):
        return self._name

    @property
    def dtype(self):
        return self._dtype

    @property
    def dist(self):
        return self._dist

    def is_observed(self):
        return self._observation is not None

    @property
    def tensor(self):
        """Observation if observed, else (cached) samples (bn.py:164-175)."""
        if self._observation is not None:
            return self._observation
        elif not hasattr(self, "_samples"):
            self._samples = self._dist.sample(n_samples=self._n_samples)
        return self._samples

    @property
    def shape(self):
        return self.tensor.shape

    def get_shape(self):
        return self.shape

    @property
    def cond_log_p(self):
        """log p(value | parents), cached (bn.py:195-204)."""
        if not hasattr(self, "_cond_log_p"):
            self._cond_log_p = self._dist.log_prob(self.tensor)
        return self._cond_log_p

    # -- tensor-like behaviour (TensorArithmeticMixin, utils.py:18-174) -----
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            if isinstance(a, StochasticTensor):
                return a.tensor
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(x) for x in a)
            return a
        kwargs = kwargs or {}
        return func(*unwrap(args), **{k: unwrap(v) for k, v in kwargs.items()})

    def __add__(self, o): return self.tensor + _val(o)
    def __radd__(self, o): return _val(o) + self.tensor
    def __sub__(self, o): return self.tensor - _val(o)
    def __rsub__(self, o): return _val(o) - self.tensor
    def __mul__(self, o): return self.tensor * _val(o)
    def __rmul__(self, o): return _val(o) * self.tensor
    def __truediv__(self, o): return self.tensor / _val(o)
    def __rtruediv__(self, o): return _val(o) / self.tensor
    def __pow__(self, o): return self.tensor ** _val(o)
    def __rpow__(self, o): return _val(o) ** self.tensor
    def __matmul__(self, o): return self.tensor @ _val(o)
    def __rmatmul__(self, o): return _val(o) @ self.tensor
    def __neg__(self): return -self.tensor
    def __abs__(self): return abs(self.tensor)
    def __getitem__(self, item): return self.tensor[item]


def _val(o):
    return o.tensor if isinstance(o, StochasticTensor) else o


class _BayesianNet(object):
    """bn.py:319-552."""

    def __init__(self):
        self._nodes = {}
        # built inside MetaBayesianNet.observe(): remember who runs the
        # builder and what it declared observed (bn.py:319-347)
        frame = active_frame()
        self._owner = frame.owner if frame is not None else None
        self._observed = frame.observed if frame is not None else {}

    @property
    def nodes(self):
        return self._nodes

    def _get_observation(self, name):
        return self._observed.get(name)

    def stochastic(self, name, dist, **kwargs):
        """Add a stochastic node (bn.py:348-371)."""
        if name in self._nodes:
            raise ValueError(
                "There exists a node with name '{}' in the {}. Names should "
                "be unique.".format(name, BayesianNet.__name__))
        if hasattr(self, "_log_joint_cache"):
            del self._log_joint_cache
        node = StochasticTensor(
            self, name, dist, observation=self._get_observation(name),
            **kwargs)
        self._nodes[name] = node
        return node

    def deterministic(self, name, input_tensor):
        """Add a named deterministic node (bn.py:373-385)."""
        # (a symbolic latent expression stays one: nothing is computed until
        # somebody uses the node's value)
        input_tensor = as_tensor(input_tensor, keep_symbolic=True)
        self._nodes[name] = input_tensor
        return input_tensor

    def _resolve(self, name_or_names, stochastic_only, pick):
        """Shared lookup of `get` / `cond_log_prob` (bn.py:420-452): one name
        -> one result, any other iterable of names -> a list; the error
        classes and messages are the reference's (bn.py:386-403)."""
        single = isinstance(name_or_names, str)
        picked = []
        for name in ((name_or_names,) if single else tuple(name_or_names)):
            if not isinstance(name, str):
                raise TypeError(
                    "Expected string in `name_or_names`, got {} of type {}."
                    .format(repr(name), type(name)))
            node = self._nodes.get(name, self)     # self: "absent" marker
            if node is self:
                raise ValueError("There isn't a node named '{}' in the {}."
                                 .format(name, BayesianNet.__name__))
            if stochastic_only and not isinstance(node, StochasticTensor):
                raise ValueError(
                    "Node '{}' is deterministic (input or output)."
                    .format(name))
            picked.append(pick(node))
        return picked[0] if single else picked

    def get(self, name_or_names):
        """Node(s) by name (bn.py:420-435)."""
        return self._resolve(name_or_names, False, lambda node: node)

    def cond_log_prob(self, name_or_names):
        """Conditional log-density of stochastic node(s) (bn.py:437-452)."""
        return self._resolve(name_or_names, True,
                             lambda node: node.cond_log_p)

    def _log_joint(self):
        """The owner's `log_joint(bn)` if one was assigned, else the sum of
        every stochastic node's conditional log-density (bn.py:454-465)."""
        custom = None if self._owner is None else self._owner.log_joint
        if custom is None:
            return sum(node.cond_log_p for node in self._nodes.values()
                       if isinstance(node, StochasticTensor))
        if not callable(custom):
            raise TypeError(
                "{}.log_joint is set to a non-callable instance: {}"
                .format(type(self._owner).__name__, repr(custom)))
        return custom(self)

    def log_joint(self):
        """Evaluated once per BayesianNet (bn.py:467-478); adding a node
        invalidates it."""
        if not hasattr(self, "_log_joint_cache"):
            self._log_joint_cache = self._log_joint()
        return self._log_joint_cache

    def __getitem__(self, name):
        return self._resolve((name,), False, lambda node: node)[0]



# -- node factories ----------------------------------------------------------
# `bn.normal(name, ...)`, `bn.bernoulli(name, ...)` ... (reference bn.py:556-965)
# are generated from this table: distribution class, then the parameters in
# the reference's positional order with their defaults (_REQUIRED = no
# default).  `n_samples` goes to the node (how many samples an unobserved node
# draws), everything else to the distribution's constructor; unknown keyword
# options are passed to both, as the reference does.
_REQUIRED = object()
_COMMON = (('n_samples', None), ('group_ndims', 0))
_NODE_FACTORIES = {
    'normal': ('Normal', (
        ('mean', 0.), ('_sentinel', None), ('std', None), ('logstd', None),
        ('group_ndims', 0), ('n_samples', None), ('is_reparameterized', True),
        ('check_numerics', False)), 'bn.py:556-590'),
    'laplace': ('Laplace', (
        ('loc', _REQUIRED), ('scale', _REQUIRED)) + _COMMON + (
        ('is_reparameterized', True), ('check_numerics', False)),
        'bn.py:1041-1068'),
    'gamma': ('Gamma', (
        ('alpha', _REQUIRED), ('beta', _REQUIRED)) + _COMMON + (
        ('check_numerics', False),), 'bn.py:713-738'),
    'inverse_gamma': ('InverseGamma', (
        ('alpha', _REQUIRED), ('beta', _REQUIRED)) + _COMMON + (
        ('check_numerics', False),), 'bn.py:1012-1039'),
    'beta': ('Beta', (
        ('alpha', _REQUIRED), ('beta', _REQUIRED)) + _COMMON + (
        ('check_numerics', False),), 'bn.py:740-765'),
    'bernoulli': ('Bernoulli', (
        ('logits', _REQUIRED),) + _COMMON + (('dtype', torch.int32),),
        'bn.py:628-654'),
    'categorical': ('Categorical', (
        ('logits', _REQUIRED),) + _COMMON + (('dtype', torch.int32),),
        'bn.py:656-682'),
    'unnormalized_multinomial': ('UnnormalizedMultinomial', (
        ('logits', _REQUIRED), ('normalize_logits', True), ('group_ndims', 0),
        ('dtype', torch.int32)), 'bn.py:938-965'),
    'multivariate_normal_cholesky': ('MultivariateNormalCholesky', (
        ('mean', _REQUIRED), ('cov_tril', _REQUIRED)) + _COMMON + (
        ('is_reparameterized', True), ('check_numerics', False)),
        'bn.py:840-870'),
}
_ALIASES = {'discrete': 'categorical',
            'bag_of_categoricals': 'unnormalized_multinomial'}


def _make_factory(method, dist_name, params, cite):
    order = [k for k, _ in params]

    def add_node(self, name, *args, **options):
        if len(args) > len(order):
            raise TypeError("%s() takes at most %d positional arguments (%d "
                            "given)" % (method, len(order) + 1, len(args) + 1))
        given = dict(zip(order, args))
        for k in list(options):
            if k in given:
                raise TypeError("%s() got multiple values for argument '%s'"
                                % (method, k))
        extras = {k: v for k, v in options.items() if k not in order}
        given.update({k: v for k, v in options.items() if k in order})
        for k, default in params:
            if k not in given:
                if default is _REQUIRED:
                    raise TypeError("%s() missing required argument: '%s'"
                                    % (method, k))
                given[k] = default
        node_opts = dict(extras)
        if 'n_samples' in given:
            node_opts['n_samples'] = given.pop('n_samples')
        dist = getattr(distributions, dist_name)(**dict(given, **extras))
        return self.stochastic(name, dist, **node_opts)
    add_node.__name__ = method
    add_node.__doc__ = "Add a %s node (reference %s)." % (dist_name, cite)
    return add_node


for _m, (_d, _p, _c) in _NODE_FACTORIES.items():
    setattr(_BayesianNet, _m, _make_factory(_m, _d, _p, _c))
for _alias, _m in _ALIASES.items():
    setattr(_BayesianNet, _alias, getattr(_BayesianNet, _m))


class BayesianNet(_BayesianNet):
    """bn.py:481-520 (the deprecated context-manager / `observed=` /
    `query()` API of :1191-1249 is out of scope)."""

    def __init__(self, observed=None):
        if observed is not None:
            raise NotImplementedError(
                "BayesianNet(observed=...) is the deprecated 0.3 API; use "
                "@meta_bayesian_net and .observe(**observed).")
        super(BayesianNet, self).__init__()
