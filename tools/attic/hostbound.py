"""Is the single-stream rate host- or GPU-bound?  Enqueue time vs completion time of N solves."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
model = synthetic.make_model("gradp")
grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
with psm_amd.GridSurrogate(model, 256, 256) as sur:
    d_in, d_out = DeviceArray(grid), DeviceArray(shape=(256, 256, 2))
    for i in range(300): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize()
    for N in (20, 100, 1000, 4000):
        t0 = time.perf_counter()
        for i in range(N): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        t1 = time.perf_counter()
        sur.synchronize()
        t2 = time.perf_counter()
        print(f"N={N:5d}: enqueue {1e6*(t1-t0)/N:6.1f} us/solve   total {1e6*(t2-t0)/N:6.1f} us/solve")
