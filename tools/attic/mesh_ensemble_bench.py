"""Solver boundary with several independent cases on one GPU (one handle per case): synchronous psm_solve calls one after
the other against psm_solve_begin on all cases followed by psm_solve_end on all (one thread)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import cases
from psm_amd import SolverModule
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = 1000
array, top, obst, model, maxs = cases.build_mesh_case()
mods, cells, outs = [], [], []
for k in range(K):
    a = np.ascontiguousarray(cases.build_mesh_case(step=k)[0], np.float64)
    sm = SolverModule(model, maxs, geometry="native")
    sm.init_func(a, top, obst)
    o = np.empty(a.shape[0])
    sm.pin(a, o)
    mods.append(sm); cells.append(a); outs.append(o)
for _ in range(30):
    for sm, a, o in zip(mods, cells, outs): sm.py_func(a, out=o)
t0 = time.perf_counter()
for _ in range(N):
    for sm, a, o in zip(mods, cells, outs): sm.py_func(a, out=o)
ds = (time.perf_counter() - t0) / (N * K)
t0 = time.perf_counter()
for _ in range(N):
    for sm, a, o in zip(mods, cells, outs): sm.py_func_begin(a, out=o)
    for sm in mods: sm.py_func_end()
da = (time.perf_counter() - t0) / (N * K)
print(f"{K} cases ({cells[0].shape[0]} cells each, registered buffers): synchronous {ds*1e6:6.1f} us per solve = {1/ds:7.0f} solves/s ; "
      f"begin-all / end-all {da*1e6:6.1f} us per solve = {1/da:7.0f} solves/s")
