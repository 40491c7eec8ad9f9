"""psm_solve on the reference's shipped case shape (bench.py's shipped_case leg), 300 calls -- for
    rocprofv3 --kernel-trace --stats -- python3 tools/attic/shipped_trace.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from psm_amd import SolverModule, synthetic
W, maxs4, maxs_pca = cases.real_chapter5_weights()
model = synthetic.make_model("chapter5", p_in=45, p_out=48, weights=W)
model.in_a, model.out_a = float(maxs_pca[0]), float(maxs_pca[1])
array, top, obst = synthetic.shipped_case_mesh()
sm = SolverModule(model, tuple(float(v) for v in maxs4), geometry="native")
sm.init_func(array, top, obst)
cells, p = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
sm.pin(cells, p)
for _ in range(20): sm.py_func(cells, out=p)
per = []
for _ in range(int(os.environ.get("PSM_SHIPPED_CALLS", "300"))):
    t0 = time.perf_counter(); sm.py_func(cells, out=p); per.append(time.perf_counter() - t0)
per.sort()
print(f"{per[len(per) // 2] * 1e6:.1f} us per call (median of {len(per)}; p10 {per[len(per) // 10] * 1e6:.1f}, p90 {per[len(per) * 9 // 10] * 1e6:.1f}); grid {sm._sur.ny} x {sm._sur.nx}, {sm._sur.B} blocks")
sm.unpin()
