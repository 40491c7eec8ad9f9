"""psm_solve on a mesh of the reference's shipped case size (400 x 3000 grid, 104 blocks): geometry bound by psm_set_geometry
(two-launch end for > 64 blocks) against the general path (PSM_NO_BIND=1), and the time per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from psm_amd import SolverModule, synthetic
W, maxs4, maxs_pca = cases.real_chapter5_weights()
model = synthetic.make_model("chapter5", p_in=45, p_out=48, weights=W)
model.in_a, model.out_a = float(maxs_pca[0]), float(maxs_pca[1])
array, top, obst = synthetic.channel_mesh(Lx=15.0, Ly=2.0, h=0.012, cx=3.0, R=0.25)
res = {}
for mode in ("bound", "general"):
    if mode == "general": os.environ["PSM_NO_BIND"] = "1"
    sm = SolverModule(model, tuple(float(v) for v in maxs4))
    t0 = time.perf_counter(); sm.init_func(array, top, obst); t_init = time.perf_counter() - t0
    cells, out = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
    sm.pin(cells, out)
    for _ in range(20): sm.py_func(cells, out=out)
    t0 = time.perf_counter()
    for _ in range(200): sm.py_func(cells, out=out)
    dt = (time.perf_counter() - t0) / 200
    res[mode] = out.copy()
    print(f"{mode:8s}: cells {array.shape[0]}, grid {sm._sur.ny}x{sm._sur.nx}, blocks {sm._sur.B}, bound={sm._sur.geometry_bound}, "
          f"init_func {t_init:.1f} s, psm_solve {dt*1e6:7.1f} us per call", flush=True)
    sm.unpin()
d = np.abs(res["bound"] - res["general"]).max() / np.abs(res["general"]).max()
print("max |p_bound - p_general| / max|p| =", d)
assert d < 5e-5
