"""MI355X-native pressure-surrogate inference path.

The directory name carries hyphens (it is fixed by the project layout), so the
package is imported through the root-level alias module ``psm_amd``.
"""
from . import formats, synthetic  # noqa: F401

__all__ = ["formats", "synthetic"]
