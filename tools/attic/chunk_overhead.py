"""Wall time of a chunk of n back-to-back bound solves between two device synchronisations, for several n: per solve and the fixed part
(fill + drain + synchronise) from a straight-line fit -- what the driver's K = 20 run sees against K = 2000.
    python tools/attic/chunk_overhead.py [workload=config1]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
wl = sys.argv[1] if len(sys.argv) > 1 else "config1"
variant, NY, NX, NC, prec, _ = bench.WORKLOADS[wl]
model = synthetic.make_model(variant)
g = synthetic.channel_grid(NY, NX, seed=1).astype(np.float32)[None]
with psm_amd.GridSurrogate(model, NY, NX, max_cases=1, precision=prec) as sur:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(1, NY, NX, model.c_out))
    assert sur.bind_geometry(d_in.ptr, on_device=True, n_cases=1)
    for i in range(500): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize()
    ns, med = [5, 10, 20, 50, 100, 200, 500, 2000], []
    for n in ns:
        ts = []
        for rep in range(60 if n <= 500 else 10):
            sur.synchronize(); t0 = time.perf_counter()
            for i in range(n): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
            sur.synchronize(); ts.append(time.perf_counter() - t0)
        med.append(np.median(ts))
        t_issue = []
        for rep in range(20):
            sur.synchronize(); t0 = time.perf_counter()
            for i in range(n): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
            t_issue.append(time.perf_counter() - t0); sur.synchronize()
        print(f"n = {n:5d}: {1e6 * med[-1]:9.1f} us per chunk = {1e6 * med[-1] / n:6.2f} us per solve ({n / med[-1]:7.0f} solves/s); host issue alone {1e6 * np.median(t_issue) / n:6.2f} us per solve")
    a, b = np.polyfit(ns, med, 1)
    print(f"fit: {1e6 * a:.2f} us per solve + {1e6 * b:.1f} us per chunk")
