"""MI355X-native pressure-surrogate inference path.

The directory name carries hyphens (it is fixed by the project layout), so the
package is imported through the root-level alias module ``psm_amd``.
"""
# (The host-buffer ring wants GPU_MAX_HW_QUEUES=16 in the environment of the HOST PROGRAM before its first GPU call -- see
# INTEGRATION.md; neither this package nor the library sets it.)
from . import _lib, dist, formats, geometry, hostinfo, surrogate, synthetic, unet  # noqa: F401
from .unet import UNetSurrogate  # noqa: F401
from .surrogate import (Evaluation, EvaluationGradP, EvaluationPoisson, GridSurrogate, SolverModule, call_SM_main,  # noqa: F401
                        call_SM_main_Poisson, error_metrics, main_gradP)
from .synthetic import SurrogateModel  # noqa: F401

__all__ = ["formats", "synthetic", "surrogate", "_lib", "dist", "GridSurrogate", "Evaluation", "EvaluationGradP", "EvaluationPoisson", "call_SM_main", "call_SM_main_Poisson", "main_gradP", "error_metrics",
           "SolverModule", "SurrogateModel"]
