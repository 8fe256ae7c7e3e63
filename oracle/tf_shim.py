"""A minimal eager stand-in for the TensorFlow-1 API surface that
/root/reference/zhusuan/hmc.py (and zhusuan/utils.py) touch, backed by
float32 torch-CPU tensors.  TEST INFRASTRUCTURE (see oracle/__init__.py):
only oracle/make_golden_hmc.py and oracle/make_golden_sgmcmc.py import it, in
the build container, to run the reference's OWN hmc.py / sgmcmc.py --
unmodified, loaded from /root/reference by file path -- and record golden
traces (tests/golden/{hmc,sgmcmc}_reference_traces.npz) that pin
oracle/hmc_ref.py and oracle/sgmcmc_ref.py.  TensorFlow itself is not installable here
(requirements-dev.txt:2 "tensorflow>=1.13.0", no wheel, no network).

How a TF-1 *graph* maps onto eager execution:

* `HMC.__init__` runs once and creates the sampler's Variables.
  `HMC.sample(...)` -- which in TensorFlow only BUILDS the graph that
  `sess.run(sample_op)` later executes -- is called once PER ITERATION here:
  every op executes eagerly in program order, so one call is one execution
  of the graph.  Variables created inside `sample()` (the EWMV state,
  hmc.py:285-286 -> :118-123) are replayed from a creation-order store on
  every call after the first (`begin_run`), i.e. they persist like graph
  variables.
* `tf.cond` runs only the taken branch, `tf.while_loop` is a Python loop,
  `tf.control_dependencies` / `tf.name_scope` are no-ops (program order is
  the order the dependencies request), `tf.gradients` is torch autograd of
  sum(ys), placeholders are objects whose value is fed before each run.
* `tf.random_normal` / `tf.random_uniform` draw from the stream the harness
  installs (`set_random_source`): the Philox mapping of oracle/philox.py, so
  the reference code, the oracle and the device see identical numbers.

Numerics: float32 torch-CPU kernels stand in for TensorFlow's Eigen kernels
(same IEEE arithmetic; reductions and exp/log/sqrt may differ in the last
bit).  What the traces pin is the reference's control flow and update
equations, executed by the reference's own code.
"""
import contextlib
import sys
import types

import numpy as np
import torch

float16 = torch.float16
float32 = torch.float32
float64 = torch.float64
int16 = torch.int16
int32 = torch.int32
int64 = torch.int64
Tensor = torch.Tensor
bool = torch.bool  # noqa: A001  (tf.bool)

_py_bool = __builtins__['bool'] if isinstance(__builtins__, dict) \
    else __builtins__.bool


class InvalidArgumentError(ArithmeticError):
    """tf.errors.InvalidArgumentError (raised by check_numerics)."""


class TensorShape(object):
    """Static shapes are always fully known in eager execution; `dims=None`
    (unknown rank) exists only so that `if not shape` keeps TensorFlow's
    meaning (hmc.py:437)."""

    def __init__(self, dims=()):
        if isinstance(dims, TensorShape):
            dims = dims.dims
        self.dims = None if dims is None else [
            None if d is None else int(d) for d in dims]

    def __len__(self):
        return len(self.dims)

    def __bool__(self):
        return self.dims is not None

    __nonzero__ = __bool__

    def __iter__(self):
        return iter(self.dims)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return TensorShape(self.dims[i])
        return self.dims[i]

    def __eq__(self, other):
        try:
            return list(self.dims) == list(TensorShape(other).dims)
        except TypeError:
            return NotImplemented

    def __ne__(self, other):
        return not self == other

    @property
    def ndims(self):
        return None if self.dims is None else len(self.dims)

    def is_fully_defined(self):
        return self.dims is not None and all(d is not None for d in self.dims)

    def concatenate(self, other):
        return TensorShape(self.dims + list(TensorShape(other).dims))

    def as_list(self):
        return list(self.dims)

    def __repr__(self):
        return 'TensorShape(%r)' % (self.dims,)

    __str__ = lambda self: str(tuple(self.dims))


def broadcast_static_shape(shape_x, shape_y):
    a, b = TensorShape(shape_x).as_list(), TensorShape(shape_y).as_list()
    n = max(len(a), len(b))
    a, b = [1] * (n - len(a)) + a, [1] * (n - len(b)) + b
    out = []
    for x, y in zip(a, b):
        if x == 1:
            out.append(y)
        elif y == 1 or x == y:
            out.append(x)
        else:
            raise ValueError('Incompatible shapes for broadcasting: %s and %s'
                             % (shape_x, shape_y))
    return TensorShape(out)


def broadcast_dynamic_shape(shape_x, shape_y):
    return torch.tensor(broadcast_static_shape(
        _shape_list(shape_x), _shape_list(shape_y)).as_list(),
        dtype=torch.int32)


def _get_shape(t):
    return TensorShape(tuple(t.shape))


# tensors are plain torch tensors; give them the one TF method hmc.py calls
torch.Tensor.get_shape = _get_shape
torch.Tensor.set_shape = lambda self, shape: None


# ---- variables with graph-like persistence --------------------------------
_VARS = []
_CURSOR = [None]      # None = create mode; int = replay index


def begin_run(mark):
    """Call before every re-execution of `sample()` after the first: the
    Variables it creates are taken from the store starting at `mark`."""
    _CURSOR[0] = mark


def variable_mark():
    return len(_VARS)


def end_replay():
    _CURSOR[0] = None


class Variable(object):
    def __new__(cls, initial_value=None, name=None, trainable=False,
                dtype=None, **kw):
        if _CURSOR[0] is not None:
            v = _VARS[_CURSOR[0]]
            _CURSOR[0] += 1
            return v
        v = object.__new__(cls)
        v._init(initial_value, name, dtype)
        _VARS.append(v)
        return v

    def __init__(self, *a, **kw):
        pass

    def _init(self, initial_value, name, dtype):
        self.name = name
        self.value = _leaf(convert_to_tensor(initial_value, dtype=dtype))

    # -- TF Variable API used by hmc.py ------------------------------------
    def assign(self, value):
        self.value = _leaf(convert_to_tensor(value).to(self.value.dtype))
        return self.value

    def assign_add(self, delta):
        return self.assign(self.value.detach() + convert_to_tensor(delta).detach())

    def get_shape(self):
        return TensorShape(tuple(self.value.shape))

    @property
    def shape(self):
        return self.value.shape

    def numpy(self):
        return self.value.detach().numpy().copy()

    # arithmetic delegates to the current value tensor
    def __add__(self, o): return self.value + _t(o)
    def __radd__(self, o): return _t(o) + self.value
    def __sub__(self, o): return self.value - _t(o)
    def __rsub__(self, o): return _t(o) - self.value
    def __mul__(self, o): return self.value * _t(o)
    def __rmul__(self, o): return _t(o) * self.value
    def __truediv__(self, o): return self.value / _t(o)
    def __rtruediv__(self, o): return _t(o) / self.value
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return -self.value
    def __pow__(self, o): return self.value ** _t(o)
    def __mod__(self, o): return torch.remainder(self.value, _t(o))
    def __lt__(self, o): return self.value < _t(o)
    def __gt__(self, o): return self.value > _t(o)


def _leaf(t):
    t = t.detach().clone()
    if t.is_floating_point():
        t.requires_grad_(True)
    return t


class Placeholder(object):
    """tf.placeholder: `.feed(v)` before a run."""

    def __init__(self, dtype, shape=None, name=None):
        self.dtype, self.name, self._v = dtype, name, None

    # value used while a harness executes graph CONSTRUCTION eagerly (an
    # unfed placeholder is legal there in TensorFlow); None = must be fed
    unfed_default = None

    def feed(self, v):
        self._v = v

    @property
    def value(self):
        v = self._v if self._v is not None else Placeholder.unfed_default
        assert v is not None, 'placeholder %s was not fed' % self.name
        return torch.as_tensor(v, dtype=self.dtype)

    # arithmetic on the fed value (evaluation.py:101-103 multiplies by the
    # temperature placeholder)
    def __add__(self, o): return self.value + _t(o)
    def __radd__(self, o): return _t(o) + self.value
    def __sub__(self, o): return self.value - _t(o)
    def __rsub__(self, o): return _t(o) - self.value
    def __mul__(self, o): return self.value * _t(o)
    def __rmul__(self, o): return _t(o) * self.value
    def __hash__(self): return id(self)


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


_CONVERTERS = []     # tf.register_tensor_conversion_function


def register_tensor_conversion_function(base_type, conversion_func,
                                        priority=100):
    _CONVERTERS.append((base_type, conversion_func))


def _t(x):
    if isinstance(x, Variable):
        return x.value
    if isinstance(x, Placeholder):
        return x.value
    for base, fn in _CONVERTERS:
        if isinstance(x, base):
            return fn(x)
    return x


def convert_to_tensor(value, dtype=None, name=None):
    if isinstance(value, Placeholder):
        return value                     # resolved where it is consumed
    if isinstance(value, Variable):
        value = value.value
    for base, fn in _CONVERTERS:
        if isinstance(value, base):
            value = fn(value)
    if isinstance(value, TensorShape):
        value = value.as_list()
    if isinstance(value, torch.Tensor):
        return value if dtype is None or value.dtype == dtype else value.to(dtype)
    if isinstance(value, np.ndarray):
        t = torch.from_numpy(np.array(value))
        return t if dtype is None else t.to(dtype)
    if dtype is None:
        dtype = torch.float32 if isinstance(value, float) else (
            torch.bool if isinstance(value, _py_bool) else (
                torch.int32 if isinstance(value, int) else torch.float32))
    return torch.tensor(value, dtype=dtype)


def constant(value, dtype=None, name=None, shape=None):
    return convert_to_tensor(value, dtype=dtype)


def cast(x, dtype, name=None):
    x = _t(x)
    if not isinstance(x, torch.Tensor):      # a Python number (tf.shape(h)[2])
        return torch.tensor(x, dtype=dtype)
    return x.to(dtype)


def identity(x, name=None):
    return _t(x)


def stop_gradient(x, name=None):
    if isinstance(x, (list, tuple)):
        return [stop_gradient(i) for i in x]
    return _t(x).detach()


def assign(ref, value, name=None):
    return ref.assign(value)


def group(*ops, **kw):
    return None


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield


@contextlib.contextmanager
def control_dependencies(deps):
    yield


def _pred(p):
    p = _t(p)
    if isinstance(p, torch.Tensor):
        return _py_bool(p.item())
    return _py_bool(p)


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None):
    true_fn = true_fn or fn1
    false_fn = false_fn or fn2
    return true_fn() if _pred(pred) else false_fn()


def while_loop(cond, body, loop_vars, back_prop=True, parallel_iterations=10,
               **kw):
    lv = list(loop_vars)
    while _pred(cond(*lv)):
        lv = list(body(*lv))
    return lv


# ---- element-wise / reductions ---------------------------------------------
def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, torch.Tensor):
        axis = axis.tolist()
    if isinstance(axis, (list, tuple)):
        return tuple(int(a) for a in axis)
    return int(axis)


def reduce_sum(x, axis=None, keepdims=False, name=None, **kw):
    x = _t(x)
    ax = _axes(axis)
    if ax is None:
        return x.sum()
    if ax == ():
        return x
    return x.sum(dim=ax, keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False, name=None, **kw):
    x = _t(x)
    ax = _axes(axis)
    if ax is None:
        return x.mean()
    if ax == ():
        return x
    return x.mean(dim=ax, keepdim=keepdims)


def add_n(xs, name=None):
    out = _t(xs[0])
    for x in xs[1:]:
        out = out + _t(x)
    return out


def square(x, name=None): return _t(x) * _t(x)
def sqrt(x, name=None): return torch.sqrt(_t(x))
def exp(x, name=None): return torch.exp(_t(x))
def log(x, name=None): return torch.log(_t(x))
def pow(x, y, name=None): return torch.pow(_tt(x), _tt(y))  # noqa: A001
def minimum(x, y, name=None): return torch.minimum(_tt(x), _tt(y))
def less(x, y, name=None): return _tt(x) < _tt(y)
def equal(x, y, name=None): return _tt(x) == _tt(y)
def logical_and(x, y, name=None): return torch.logical_and(_tt(x), _tt(y))
def logical_or(x, y, name=None): return torch.logical_or(_tt(x), _tt(y))
def logical_xor(x, y, name=None): return torch.logical_xor(_tt(x), _tt(y))
def logical_not(x, name=None): return torch.logical_not(_tt(x))
def is_finite(x, name=None): return torch.isfinite(_t(x))
def abs(x, name=None): return torch.abs(_t(x))  # noqa: A001
def negative(x, name=None): return -_t(x)
def mod(x, y, name=None): return torch.remainder(_tt(x), _tt(y))
def add(x, y, name=None): return _tt(x) + _tt(y)
def subtract(x, y, name=None): return _tt(x) - _tt(y)
def multiply(x, y, name=None): return _tt(x) * _tt(y)
def divide(x, y, name=None): return _tt(x) / _tt(y)
truediv = divide


def _tt(x):
    x = _t(x)
    return x if isinstance(x, torch.Tensor) else convert_to_tensor(x)


def where(condition, x=None, y=None, name=None):
    return torch.where(_tt(condition), _tt(x), _tt(y))


def expand_dims(x, axis, name=None):
    return _t(x).unsqueeze(axis)


def shape(x, name=None):
    return tuple(_t(x).shape)


def _shape_list(s):
    if isinstance(s, TensorShape):
        return s.as_list()
    if isinstance(s, torch.Tensor):
        return [int(v) for v in s.tolist()]
    return [int(v) for v in s]


def zeros(shape=None, dtype=torch.float32, name=None):
    return torch.zeros(_shape_list(shape), dtype=dtype)


def ones(shape=None, dtype=torch.float32, name=None):
    return torch.ones(_shape_list(shape), dtype=dtype)


def zeros_like(x, dtype=None, name=None):
    return torch.zeros_like(_t(x), dtype=dtype)


def ones_like(x, dtype=None, name=None):
    return torch.ones_like(_t(x), dtype=dtype)


def range(*a, **kw):  # noqa: A001
    return torch.arange(*a)


def check_numerics(x, message, name=None):
    x = _t(x)
    if not _py_bool(torch.isfinite(x).all().item()):
        raise InvalidArgumentError(message + ' : Tensor had Inf or NaN values')
    return x


def gradients(ys, xs, **kw):
    """d sum(ys) / d xs (tf.gradients semantics), xs Variables or tensors."""
    single = not isinstance(xs, (list, tuple))
    xl = [xs] if single else list(xs)
    leaves = [_t(x) for x in xl]
    y = _t(ys)
    gs = torch.autograd.grad(y.sum(), leaves, allow_unused=True,
                             retain_graph=True)
    gs = [torch.zeros_like(l) if g is None else g.detach()
          for g, l in zip(gs, leaves)]
    return gs


# ---- random ops: the harness installs the source ----------------------------
_RANDOM = {'normal': None, 'uniform': None}


def set_random_source(normal_fn, uniform_fn):
    """normal_fn(shape) / uniform_fn(shape) -> float32 numpy arrays; called
    in program order (one random_normal per latent in random_momentum,
    hmc.py:21-23; one random_uniform in the MH block, :485)."""
    _RANDOM['normal'], _RANDOM['uniform'] = normal_fn, uniform_fn


def random_normal(shape, mean=0.0, stddev=1.0, dtype=torch.float32, seed=None,
                  name=None):
    z = torch.from_numpy(np.ascontiguousarray(
        _RANDOM['normal'](tuple(_shape_list(shape))), dtype=np.float32))
    return z * stddev + mean


def random_uniform(shape, minval=0, maxval=None, dtype=torch.float32,
                   seed=None, name=None):
    return torch.from_numpy(np.ascontiguousarray(
        _RANDOM['uniform'](tuple(_shape_list(shape))), dtype=np.float32))


# ---- the rest of the surface the model layer touches ------------------------
# (zhusuan/framework/{bn,meta_bn,utils}.py, distributions/{base,utils,
# univariate}.py, evaluation.py -- loaded unmodified by
# oracle/make_golden_model.py)
@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, reuse=None, **kw):
    yield


def make_template(name, func, **kw):
    return func


def squeeze(x, axis=None, name=None):
    x = _t(x)
    return x.squeeze() if axis is None else x.squeeze(_axes(axis))


def reduce_prod(x, axis=None, keepdims=False, name=None, **kw):
    x = _t(x)
    ax = _axes(axis)
    if ax is None:
        return x.prod()
    if ax == ():
        return x
    for a in sorted([a % x.dim() for a in (ax if isinstance(ax, tuple)
                                           else (ax,))], reverse=True):
        x = x.prod(dim=a, keepdim=keepdims)
    return x


def reduce_max(x, axis=None, keepdims=False, name=None, **kw):
    x = _t(x)
    ax = _axes(axis)
    return x.max() if ax is None else x.amax(dim=ax, keepdim=keepdims)


def reduce_all(x, axis=None, name=None, **kw):
    return _t(x).all()


def reduce_logsumexp(x, axis=None, keepdims=False, name=None, **kw):
    x = _t(x)
    ax = _axes(axis)
    if ax is None:
        ax = tuple(__builtins__['range'](x.dim()) if isinstance(
            __builtins__, dict) else __builtins__.range(x.dim()))
    return torch.logsumexp(x, dim=ax, keepdim=keepdims)


def _noop_assert(*a, **kw):
    return None


assert_rank = assert_rank_at_least = assert_greater = _noop_assert
assert_greater_equal = assert_positive = assert_non_negative = _noop_assert


def concat(values, axis, name=None):
    return torch.cat([_tt(v).reshape(-1) if _tt(v).dim() == 0 else _tt(v)
                      for v in values], dim=axis)


def stack(values, axis=0, name=None):
    return torch.stack([_tt(v) for v in values], dim=axis)


def reshape(x, s, name=None):
    return _t(x).reshape(_shape_list(s))


def transpose(x, perm=None, name=None):
    x = _t(x)
    return x.t() if perm is None else x.permute(*[int(i) for i in perm])


def rank(x, name=None):
    return _t(x).dim()


def sigmoid(x, name=None): return torch.sigmoid(_t(x))
def sign(x, name=None): return torch.sign(_t(x))
def lgamma(x, name=None): return torch.lgamma(_t(x))
def less_equal(x, y, name=None): return _tt(x) <= _tt(y)
def greater(x, y, name=None): return _tt(x) > _tt(y)
def greater_equal(x, y, name=None): return _tt(x) >= _tt(y)
def floordiv(x, y, name=None): return torch.floor_divide(_tt(x), _tt(y))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return a @ b


def tile(x, multiples, name=None):
    return _t(x).repeat(*[int(m) for m in _shape_list(multiples)])


def _softmax(logits, axis=-1, name=None):
    return torch.softmax(_t(logits), dim=axis)


def _sigmoid_cross_entropy_with_logits(labels=None, logits=None, name=None):
    # tf.nn.sigmoid_cross_entropy_with_logits:
    # max(x, 0) - x * z + log(1 + exp(-abs(x)))
    x, z = _t(logits), _t(labels)
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))


def einsum(equation, *inputs, **kw):
    return torch.einsum(equation, *[_t(x) for x in inputs])


def gather(params, indices, axis=0, name=None, **kw):
    idx = _t(indices).to(torch.int64)
    return torch.index_select(_t(params), int(axis), idx.reshape(-1)).reshape(
        tuple(_t(params).shape[:int(axis)]) + tuple(idx.shape) +
        tuple(_t(params).shape[int(axis) + 1:]))


def _sparse_softmax_cross_entropy_with_logits(labels=None, logits=None,
                                             name=None):
    # logsumexp(logits) - logits[label]
    x, k = _t(logits), _t(labels).to(torch.int64)
    return torch.logsumexp(x, dim=-1) - x.gather(-1, k.unsqueeze(-1)).squeeze(-1)


def install():
    """Register this module as `tensorflow` (only if the real one is absent)."""
    mod = sys.modules[__name__]
    if 'tensorflow' in sys.modules and sys.modules['tensorflow'] is not mod:
        raise RuntimeError('a real tensorflow is importable: use it instead')
    sys.modules['tensorflow'] = mod
    errors = types.ModuleType('tensorflow.errors')
    errors.InvalidArgumentError = InvalidArgumentError
    mod.errors = errors
    # tensorflow.python.client.session (framework/bn.py:10-11)
    mod.__path__ = []
    py = types.ModuleType('tensorflow.python')
    py.__path__ = []
    client = types.ModuleType('tensorflow.python.client')
    client.__path__ = []
    session = types.ModuleType('tensorflow.python.client.session')
    session.register_session_run_conversion_functions = \
        lambda *a, **kw: None
    client.session, py.client, mod.python = session, client, py
    sys.modules['tensorflow.python'] = py
    sys.modules['tensorflow.python.client'] = client
    sys.modules['tensorflow.python.client.session'] = session
    nn = types.ModuleType('tensorflow.nn')
    nn.softplus = lambda x, name=None: torch.nn.functional.softplus(_t(x))
    nn.softmax = _softmax
    nn.relu = lambda x, name=None: torch.relu(_t(x))
    nn.sigmoid_cross_entropy_with_logits = _sigmoid_cross_entropy_with_logits
    nn.sparse_softmax_cross_entropy_with_logits = \
        _sparse_softmax_cross_entropy_with_logits
    mod.nn = nn
    return mod
