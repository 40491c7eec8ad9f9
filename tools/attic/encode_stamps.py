import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = synthetic.make_model("deltas")
grid = synthetic.random_obstacle_cases(NC, 256, 256, seed=3).astype(np.float32)
with psm_amd.GridSurrogate(model, 256, 256, max_cases=NC) as sur:
    d_in, d_out = DeviceArray(grid), DeviceArray(shape=(NC, 256, 256, model.c_out))
    acc = []
    for it in range(30):
        for k in range(5): sur.solve_device(d_in.ptr, NC, d_out.ptr, 0)
        sur.synchronize()
        out = np.zeros(64, np.float32)
        sur._chk(sur.lib.psm_read_stage(sur.h, 6, out.ctypes.data_as(C.POINTER(C.c_float)), 64))
        acc.append(out.copy())
    a = np.array(acc[5:]); base = a[:, :1]
    med = np.median(a - base, axis=0)
    print("multiplying wave 0: entry, basis split, first barrier passed:", " ".join(f"{med[k]:6.2f}" for k in range(3)))
    print("  step done (before its barrier):", " ".join(f"{med[k]:6.2f}" for k in range(3, 32) if acc[-1][k] >= 0))
    print("staging wave: write of step j+1 done (before barrier j):", " ".join(f"{med[k]:6.2f}" for k in range(32, 64) if acc[-1][k] >= 0))
