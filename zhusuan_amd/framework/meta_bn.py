"""MetaBayesianNet: a model builder plus its call arguments, re-run under a
set of observations whenever a BayesianNet is needed.  Same surface as
reference zhusuan/framework/meta_bn.py:29-148 (`MetaBayesianNet(f, args,
kwargs, scope, reuse_variables)`, `.observe(**obs)`, assignable `.log_joint`,
the `@meta_bayesian_net` decorator) -- this is how HMC re-evaluates the joint at
a new state (hmc.py:412-416).

How a BayesianNet created inside the builder learns what is observed: a
per-thread stack of observation frames.  `observe()` pushes (owner,
observations) for the duration of the builder call; `BayesianNet()` reads the
innermost frame (`active_frame`) when it is constructed."""
import functools
import threading
from collections import namedtuple
from contextlib import contextmanager

__all__ = ['MetaBayesianNet', 'meta_bayesian_net']

ObservationFrame = namedtuple('ObservationFrame', 'owner observed')
_tls = threading.local()


def _frames():
    stack = getattr(_tls, 'frames', None)
    if stack is None:
        stack = _tls.frames = []
    return stack


@contextmanager
def _observing(owner, observed):
    stack = _frames()
    stack.append(ObservationFrame(owner, dict(observed)))
    try:
        yield
    finally:
        stack.pop()


def active_frame():
    """The innermost (owner, observed) frame of this thread, or None outside
    of any `observe()` call."""
    stack = _frames()
    return stack[-1] if stack else None


class MetaBayesianNet(object):
    """`scope` / `reuse_variables` are accepted for signature compatibility
    (meta_bn.py:49-63): there are no variable scopes here, parameters are
    plain device tensors owned by the caller; asking for reuse without a scope
    is still the reference's error."""

    def __init__(self, f, args=None, kwargs=None, scope=None,
                 reuse_variables=False):
        if reuse_variables and scope is None:
            raise ValueError("Cannot reuse tensorflow Variables when `scope` "
                             "is not provided.")
        self._builder = f
        self._call_args = tuple(args) if args else ()
        self._call_kwargs = dict(kwargs) if kwargs else {}
        self.scope = scope
        self.reuse_variables = bool(reuse_variables)
        self._joint_fn = None

    # `meta_bn.log_joint = fn(bn)` replaces the default sum over stochastic
    # nodes (meta_bn.py:69-85); None restores it
    log_joint = property(lambda self: self._joint_fn)

    @log_joint.setter
    def log_joint(self, fn):
        self._joint_fn = fn

    def observe(self, **observed):
        """Run the builder with `observed` (node name -> value) in force and
        return the BayesianNet it built (meta_bn.py:93-106)."""
        with _observing(self, observed):
            return self._builder(*self._call_args, **self._call_kwargs)


def meta_bayesian_net(scope=None, reuse_variables=False):
    """`@meta_bayesian_net(...)`: calling the decorated builder returns a
    MetaBayesianNet bound to the call's arguments instead of running it
    (meta_bn.py:109-148)."""
    def decorate(builder):
        @functools.wraps(builder)
        def bind(*args, **kwargs):
            return MetaBayesianNet(builder, args, kwargs, scope=scope,
                                   reuse_variables=reuse_variables)
        return bind
    return decorate
