from .bn import BayesianNet, StochasticTensor
from .meta_bn import MetaBayesianNet, meta_bayesian_net
from .utils import Context

__all__ = ['BayesianNet', 'StochasticTensor', 'MetaBayesianNet',
           'meta_bayesian_net', 'Context']
