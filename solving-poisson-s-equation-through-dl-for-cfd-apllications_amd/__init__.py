"""MI355X-native pressure-surrogate inference path.

The directory name carries hyphens (it is fixed by the project layout), so the
package is imported through the root-level alias module ``psm_amd``.
"""
import os as _os

# enough HIP hardware queues for the tickets of the host-buffer ring (see csrc/psm_api.cpp psm_default_hw_queues): only
# effective when this import precedes the first GPU call of the process
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _lib, dist, formats, geometry, hostinfo, surrogate, synthetic, unet  # noqa: F401
from .unet import UNetSurrogate  # noqa: F401
from .surrogate import Evaluation, EvaluationGradP, EvaluationPoisson, GridSurrogate, SolverModule, call_SM_main  # noqa: F401
from .synthetic import SurrogateModel  # noqa: F401

__all__ = ["formats", "synthetic", "surrogate", "_lib", "dist", "GridSurrogate", "Evaluation", "EvaluationGradP", "EvaluationPoisson", "call_SM_main",
           "SolverModule", "SurrogateModel"]
