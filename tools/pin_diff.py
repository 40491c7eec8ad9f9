"""psm_solve on registered (pinned) buffers against the default path over seven velocity scales: mismatching pressures per scale.
The check that isolated the one-ulp U_max difference of round 6 (profiles/r06_embed_host.txt); 0 mismatches on every scale since.
    python tools/pin_diff.py        (PSM_MESH_GRAPH=0 / PSM_NO_DIRECT_IN=1 select the other forms of the pinned path)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import cases
from psm_amd import SolverModule
array, top, obst, model, maxs = cases.build_mesh_case()
ref = SolverModule(model, maxs); ref.init_func(array, top, obst)
pin = SolverModule(model, maxs); pin.init_func(array, top, obst)
buf = np.ascontiguousarray(array, np.float64).copy(); out = np.empty(buf.shape[0])
pin.pin(buf, out)
for s in range(7):
    scale = 1.0 + 0.01 * s
    buf[:, 0] = array[:, 0] * scale; buf[:, 1] = array[:, 1] * scale
    a = buf.copy()
    p0 = ref.py_func(a).copy()
    p1 = pin.py_func(buf, out=out).copy()
    d = p0 != p1
    um = np.sqrt(np.max(np.square(a[:, 0]) + np.square(a[:, 1])))
    print(s, "mismatch", int(d.sum()), "of", d.size, "umax", repr(um), "max rel", float(np.max(np.abs(p0 - p1) / np.maximum(np.abs(p0), 1e-300))))
