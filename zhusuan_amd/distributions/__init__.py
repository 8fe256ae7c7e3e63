from .base import Distribution
from .univariate import (Normal, Bernoulli, Categorical, Discrete,
                         LinearLogits, linear_logits)
from .multivariate import UnnormalizedMultinomial, BagofCategoricals

__all__ = ['Distribution', 'Normal', 'Bernoulli', 'Categorical', 'Discrete',
           'UnnormalizedMultinomial', 'BagofCategoricals', 'LinearLogits',
           'linear_logits']
