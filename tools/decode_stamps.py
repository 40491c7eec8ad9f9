"""In-kernel stamps of the case-batch decode + paste (diagnostic build: make -C <csrc> stamps; PSM_LIB=.../libpsm_hip_stamps.so):
workgroup (0, 0), thread 0 -- entry, basis split done, then per row chunk: loop top, tile in LDS (barrier passed), next chunk requested,
MFMAs issued, stores issued.  us after the kernel's entry stamp, median of 25 samples.
    PSM_LIB=$PWD/solving-..._amd/libpsm_hip_stamps.so python tools/decode_stamps.py [cases]"""
import ctypes as C, sys
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
    assert sur.bind_geometry(d_in.ptr, on_device=True, n_cases=NC)
    acc = []
    for it in range(30):
        for k in range(5): sur.solve_device(d_in.ptr, NC, d_out.ptr, 0)
        sur.synchronize()
        out = np.zeros(64, np.float32)
        sur._chk(sur.lib.psm_read_stage(sur.h, 6, out.ctypes.data_as(C.POINTER(C.c_float)), 64))
        acc.append(out.copy())
    a = np.array(acc[5:])
    med = np.median(a - a[:, :1], axis=0)
    print(f"{NC} cases: entry 0.00, basis split done {med[1]:.2f}")
    for c in range(7):
        ks = [2 + 5 * c + j for j in range(5)]
        if ks[-1] >= 40 or acc[-1][ks[0]] < 0: break
        print(f"  chunk {c}: top {med[ks[0]]:6.2f} | tile in LDS {med[ks[1]]:6.2f} | next requested {med[ks[2]]:6.2f} | MFMAs issued {med[ks[3]]:6.2f} | stores issued {med[ks[4]]:6.2f}")
