"""psm_solve with registered buffers, 300 calls (for rocprofv3 --kernel-trace --memory-copy-trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from psm_amd import SolverModule
array, top, obst, model, maxs = cases.build_mesh_case()
sm = SolverModule(model, maxs)
sm.init_func(array, top, obst)
cells, out = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
sm.pin(cells, out)
for _ in range(300): sm.py_func(cells, out=out)
sm.unpin()
