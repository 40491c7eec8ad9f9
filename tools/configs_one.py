"""A few steps of the deltas path with N cases per call (for counter collection)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synthetic.make_model("deltas")
g = synthetic.random_obstacle_cases(n, 256, 256, seed=3).astype(np.float32)
with psm_amd.GridSurrogate(model, 256, 256, max_cases=n) as sur:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(n, 256, 256, 1))
    for i in range(30): sur.solve_device(d_in.ptr, n, d_out.ptr, 0)
    sur.synchronize()
