"""Execution plans of zhusuan_amd.hmc.HMC (one module per plan family)."""
from .base import _PlanBase, _Unsupported, _prod, _versions  # noqa: F401
from .dense import _DenseLikelihoodPlan  # noqa: F401
from .families import FAMILIES  # noqa: F401
from .fused import _FusedDiagNormalPlan, _try_fused_plan  # noqa: F401
from .generic import _GenericPlan  # noqa: F401
from .recognise import (_try_dense_likelihood_plan,  # noqa: F401
                        _try_gathered_dot_plan)
