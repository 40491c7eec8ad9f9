"""Rate of the solver boundary psm_solve (cells[N,5] float64 -> p[N] float64; PythonComm.H contract): host buffers in,
host buffers out, synchronous like py_func -- next to the NumPy oracle's py_func on the same case."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from oracle import psm_oracle as orc
from psm_amd import SolverModule
from test_oracle_golden import oracle_model
array, top, obst, model, maxs = cases.build_mesh_case()
sm = SolverModule(model, maxs)
t0 = time.perf_counter(); sm.init_func(array, top, obst); t_init = time.perf_counter() - t0
for _ in range(50): p = sm.py_func(array)
N = 2000
t0 = time.perf_counter()
for _ in range(N): p = sm.py_func(array)
dt = (time.perf_counter() - t0) / N
cells, out = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
sm.pin(cells, out)
for _ in range(50): sm.py_func(cells, out=out)
t0 = time.perf_counter()
for _ in range(N): sm.py_func(cells, out=out)
dp = (time.perf_counter() - t0) / N
print(f"psm_solve with psm_pin_buffers (direct DMA): {dp*1e6:7.1f} us per call = {1/dp:8.0f} solves/s")
sm.unpin()
geo = orc.init_geometry(array, top, obst)
om = oracle_model(model)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 5: ref = orc.py_func_mesh(array, geo, om, maxs)[0]; n += 1
dc = (time.perf_counter() - t0) / n
print(f"cells {array.shape[0]}, grid {sm._sur.ny}x{sm._sur.nx}, blocks {sm._sur.B}: init_func {t_init:.2f} s (host, SciPy qhull)")
print(f"psm_solve (py_func): {dt*1e6:7.1f} us per call = {1/dt:8.0f} solves/s ; NumPy oracle py_func {dc*1e3:6.2f} ms ({dt and dc/dt:.0f}x)")
print("max |p - oracle| / max|p| =", np.abs(p - ref).max() / np.abs(ref).max())
