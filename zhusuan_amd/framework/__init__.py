from .bn import BayesianNet, StochasticTensor
from .meta_bn import MetaBayesianNet, meta_bayesian_net

__all__ = ['BayesianNet', 'StochasticTensor', 'MetaBayesianNet',
           'meta_bayesian_net']
