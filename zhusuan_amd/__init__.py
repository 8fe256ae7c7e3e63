"""zhusuan_amd -- the zhusuan.HMC hot path rebuilt for AMD MI355X (gfx950):
Python host code over a ctypes C-ABI (include/zshmc.h, libzshmc.so) of
hand-written HIP kernels.  Same surface as `import zhusuan as zs` for the
path: zs.HMC, zs.HMCInfo, zs.BayesianNet, zs.meta_bayesian_net,
zs.distributions.{Normal, Bernoulli, Categorical, UnnormalizedMultinomial},
zs.diagnostics.effective_sample_size, and the callers next to it:
zs.AIS, zs.SGLD / PSGLD / SGHMC / SGNHT."""
from . import diagnostics, distributions, evaluation, framework
from .framework import (BayesianNet, MetaBayesianNet, StochasticTensor,
                        meta_bayesian_net)
from .distributions import (linear_class_logits, linear_logits,
                            log_mixture)
from ._ops import gathered_dot, clear_caches
from .evaluation import AIS
from .hmc import (HMC, HMCInfo, InvalidArgumentError,
                  NativePlanFallbackWarning, LikelihoodArithmeticWarning,
                  placeholder, deferred)
from .session import Session
from .sgmcmc import SGMCMC, SGLD, PSGLD, SGHMC, SGNHT
from .utils import merge_dicts, set_random_seed

__version__ = '0.1.0'

__all__ = ['SGMCMC', 'SGLD', 'PSGLD', 'SGHMC', 'SGNHT', 'AIS', 'evaluation', 'HMC', 'HMCInfo', 'InvalidArgumentError', 'NativePlanFallbackWarning', 'LikelihoodArithmeticWarning', 'placeholder', 'deferred', 'Session',
           'BayesianNet', 'MetaBayesianNet', 'StochasticTensor',
           'meta_bayesian_net', 'distributions', 'diagnostics', 'framework',
           'merge_dicts', 'set_random_seed', 'linear_logits', 'linear_class_logits', 'log_mixture', 'gathered_dot', 'clear_caches']
