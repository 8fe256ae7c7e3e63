"""MetaBayesianNet: lazily re-runs the model builder under a set of
observations.  Mirrors reference zhusuan/framework/meta_bn.py:21-148 -- this
is how HMC re-evaluates the joint at a new q (hmc.py:412-416)."""
import copy
from functools import wraps

from .utils import Context

__all__ = ['MetaBayesianNet', 'meta_bayesian_net']


class Local(Context):
    """meta_bn.py:21-26."""

    def __getattr__(self, item):
        return self.__dict__.get(item, None)

    def __setattr__(self, key, value):
        self.__dict__[key] = value


class MetaBayesianNet(object):
    """meta_bn.py:29-106.  `scope` / `reuse_variables` are accepted for
    signature compatibility; there are no TF variable scopes here, parameters
    are plain device tensors owned by the caller."""

    def __init__(self, f, args=None, kwargs=None, scope=None,
                 reuse_variables=False):
        if reuse_variables and scope is None:
            raise ValueError("Cannot reuse tensorflow Variables when `scope` "
                             "is not provided.")
        self._f = f
        self._args = copy.copy(args) if args is not None else ()
        self._kwargs = copy.copy(kwargs) if kwargs is not None else {}
        self._scope = scope
        self._reuse_variables = reuse_variables
        self._log_joint = None

    @property
    def log_joint(self):
        """The log joint function of this model; may be overwritten with a
        callable taking the BayesianNet (meta_bn.py:69-85)."""
        return self._log_joint

    @log_joint.setter
    def log_joint(self, value):
        self._log_joint = value

    def _run_with_observations(self, func, observations):
        with Local() as local_cxt:
            local_cxt.observations = observations
            local_cxt.meta_bn = self
            return func(*self._args, **self._kwargs)

    def observe(self, **kwargs):
        """Build the BayesianNet with the given observations
        (meta_bn.py:93-106)."""
        return self._run_with_observations(self._f, kwargs)


def meta_bayesian_net(scope=None, reuse_variables=False):
    """Decorator turning a BayesianNet builder into a MetaBayesianNet factory
    (meta_bn.py:109-148)."""
    def wrapper(f):
        @wraps(f)
        def _wrapped(*args, **kwargs):
            return MetaBayesianNet(f, args=args, kwargs=kwargs, scope=scope,
                                   reuse_variables=reuse_variables)
        return _wrapped
    return wrapper
