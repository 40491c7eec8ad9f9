"""PCIe-inclusive rates of the host-buffer entries (BASELINE config 1, 256x256 gradp, batch 1):
psm_solve_grid (synchronous, like py_func) and the psm_submit_grid / psm_wait_grid ring at depth 1..4,
next to the device-resident rate of bench.py."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
model = synthetic.make_model("gradp")
grids = [synthetic.channel_grid(256, 256, seed=1 + i).astype(np.float32) for i in range(4)]
N = 3000
BIND = len(sys.argv) > 1 and sys.argv[1] == "bind"     # the 4 rotated grids share one geometry: bind it (6 launches per solve)
with psm_amd.GridSurrogate(model, 256, 256) as sur:
    if BIND: assert sur.bind_geometry(grids[0]); print("geometry bound")
    d_in, d_out = DeviceArray(grids[0]), DeviceArray(shape=(256, 256, 2))
    for i in range(300): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize()
    t0 = time.perf_counter()
    for i in range(N): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize()
    dt = time.perf_counter() - t0
    print(f"device-resident, one stream      : {N/dt:9.0f} solves/s ({1e6*dt/N:6.1f} us)")
    for i in range(100): sur.solve(grids[i & 3])
    t0 = time.perf_counter()
    for i in range(N): sur.solve(grids[i & 3])
    dt = time.perf_counter() - t0
    print(f"psm_solve_grid (sync H2D+D2H)    : {N/dt:9.0f} solves/s ({1e6*dt/N:6.1f} us)")
    for depth in (1, 2, 3, 4):
        pend = []
        t0 = time.perf_counter()
        for i in range(N):
            if len(pend) == depth: sur.wait(pend.pop(0))
            pend.append(sur.submit(grids[i & 3]))
        while pend: sur.wait(pend.pop(0))
        dt = time.perf_counter() - t0
        print(f"submit/wait ring, {depth} in flight    : {N/dt:9.0f} solves/s ({1e6*dt/N:6.1f} us)")
