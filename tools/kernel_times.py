"""Per-kernel dispatch times of one BASELINE config through psm_time_kernels (the probe bench.py uses), bound path:
    python tools/kernel_times.py [workload=config1] [steps=2000]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench, psm_amd
from psm_amd import synthetic
sys.path.insert(0, 'tests')
from hipmem import DeviceArray
wl = sys.argv[1] if len(sys.argv) > 1 else "config1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
variant, NY, NX, NC, prec, _ = bench.WORKLOADS[wl]
import os
NC = int(os.environ.get("PSM_KT_CASES", NC))          # e.g. 64 cases per step on the config3 model
model = synthetic.make_model(variant)
g = synthetic.channel_grid(NY, NX, seed=1).astype(np.float32)[None] if NC == 1 else synthetic.random_obstacle_cases(NC, NY, NX, seed=3).astype(np.float32)
with psm_amd.GridSurrogate(model, NY, NX, max_cases=NC, precision=prec) as sur:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(NC, NY, NX, model.c_out))
    assert sur.bind_geometry(d_in.ptr, on_device=True, n_cases=NC)
    for rep in range(3):
        kt = bench.time_kernels(sur, d_in.ptr, NC, d_out.ptr, steps)
    tot = 0.0
    for nm, us, n in kt:
        print(f"{nm:52s} {us:7.2f} us x {n / steps:.0f}")
        tot += us * n / steps
    print(f"sum of dispatch durations per solve: {tot:.2f} us")
    import time
    for i in range(300): sur.solve_device(d_in.ptr, NC, d_out.ptr, 0)
    sur.synchronize(); t0 = time.perf_counter()
    for i in range(steps): sur.solve_device(d_in.ptr, NC, d_out.ptr, 0)
    sur.synchronize(); dt = time.perf_counter() - t0
    print(f"back-to-back: {1e6 * dt / steps:.2f} us per solve, {steps / dt:.0f} solves/s")
