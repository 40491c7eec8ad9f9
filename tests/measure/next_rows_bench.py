"""The rows SURVEY.md section 8 marks "next" (callers and post-steps either side of the hot path), timed through the host API
(host buffers in and out, synchronous) next to the CPU oracle on the same inputs: integration of (dp/dx, dp/dy) into p
(Eval_dual_Dense_onlycil.py:371-416, 592-628), ndimage.gaussian_filter of the post-steps (SM_call.py:353-363), the
pressureSM_Poisson feature image (SM_call.py:588-711) and the mesh -> grid interpolation of the evaluators."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate, synthetic


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    return (time.perf_counter() - t0) / n, out


# ---- integration of gradp
ic = cases.build_integration_case()
Ny, Nx = ic["sdfunct"].shape
cx, cy = orc.integration_center(ic["sdfunct"], ic["min_x"], ic["max_x"], ic["X0"].min(), ic["delta"])
dx, dy = (ic["max_x"] - ic["min_x"]) / (Nx - 1), (ic["max_y"] - ic["min_y"]) / (Ny - 1)
model = synthetic.make_model("gradp", p_in=8, p_out=8)
with GridSurrogate(model, Ny, Nx) as sur:
    sur.set_integration(ic["sdfunct"], cy, cx, dx, dy)
    tg, p = timeit(lambda: sur.integrate_gradp(ic["gradP"]), 300)
    tc, ref = timeit(lambda: orc.integrate_gradp(ic["gradP"], ic["sdfunct"], dx, dy, cy, cx), 5)
    print(f"integrate_gradp {Ny}x{Nx}:        GPU {tg*1e6:8.1f} us   oracle {tc*1e3:8.2f} ms   ({tc/tg:6.0f}x)   max rel diff {np.abs(p-ref).max()/np.abs(ref).max():.1e}")
    # ---- gaussian filter
    import scipy.ndimage as ndi
    f = np.random.default_rng(3).standard_normal((256, 256)).astype(np.float32)
    for sig in ((10, 10), (50, 50)):
        tg, got = timeit(lambda: sur.gaussian_filter(f, sig), 300)
        tc, ref = timeit(lambda: ndi.gaussian_filter(f.astype(np.float64), sigma=sig, order=0), 20)
        print(f"gaussian_filter 256x256 sigma {sig[0]:2d}: GPU {tg*1e6:8.1f} us   SciPy  {tc*1e3:8.2f} ms   ({tc/tg:6.0f}x)   max abs diff {np.abs(got-ref).max():.1e}")
# ---- Poisson feature image
pc = cases.build_poisson_case()
m4 = synthetic.make_model("deltas", p_in=16, p_out=16, c_in=4); m4.sdf_ch = 3
with GridSurrogate(m4, *pc["sdfunct"].shape) as sur:
    args = (pc["ux"], pc["uy"], pc["dux"], pc["duy"], pc["sdfunct"], pc["L"], pc["U"], pc["k"], pc["max_abs"])
    tg, got = timeit(lambda: sur.poisson_features(*args), 300)
    tc, ref = timeit(lambda: orc.poisson_features(*args), 20)
    ref = ref[0] if isinstance(ref, tuple) else ref
    got = got[0] if isinstance(got, tuple) else got
    print(f"poisson_features {pc['sdfunct'].shape[0]}x{pc['sdfunct'].shape[1]}:       GPU {tg*1e6:8.1f} us   oracle {tc*1e3:8.2f} ms   ({tc/tg:6.0f}x)   max abs diff {np.nanmax(np.abs(got-ref)):.1e}")
