from .base import Distribution
from .univariate import (Normal, Bernoulli, Categorical, Discrete,
                         LinearLogits, linear_logits, LinearClassLogits,
                         linear_class_logits)
from .univariate2 import Laplace, Gamma, InverseGamma, Beta
from .multivariate import (UnnormalizedMultinomial, BagofCategoricals,
                           LogMixture, log_mixture,
                           MultivariateNormalCholesky)

__all__ = ['Distribution', 'Laplace', 'Gamma', 'InverseGamma', 'Beta', 'Normal', 'Bernoulli', 'Categorical', 'Discrete',
           'UnnormalizedMultinomial', 'BagofCategoricals', 'LinearLogits',
           'LogMixture', 'log_mixture', 'MultivariateNormalCholesky',
           'linear_logits', 'LinearClassLogits', 'linear_class_logits']
