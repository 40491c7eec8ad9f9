"""In-kernel stamps of the M-tiled x6 encode (psm_encode_x6_mt_kernel; diagnostic build: make -C <csrc> stamps EXTRA=-DPSM_STAMPS_ENC,
PSM_LIB=.../libpsm_hip_stamps.so): workgroup 0, thread 0 -- entry, prologue done (first tile in LDS), then per half-slice: row tile 0's
first six MFMAs issued (= its basis planes had landed), all issued, staged + barrier; the same for row tile 1.  us after the kernel's
entry stamp, median of 25 samples.
    PSM_LIB=$PWD/solving-..._amd/libpsm_hip_stamps.so python tools/encode_stamps.py [cases]"""
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
    print(f"{NC} cases: entry 0.00, first tile staged {med[1]:.2f}, end {med[63]:.2f}")
    prev = med[1]
    for hs in range(10):
        k = 2 + 6 * hs
        if k + 5 >= 63 or acc[-1][k] < 0 or med[k] <= 0: break
        v = med[k:k + 6]
        print(f"  half-slice {hs}: tile 0  first MFMAs {v[0]:6.2f} (+{v[0]-prev:4.2f}) | all issued {v[1]:6.2f} (+{v[1]-v[0]:4.2f}) | staged+barrier {v[2]:6.2f} (+{v[2]-v[1]:4.2f})"
              f"   tile 1  first {v[3]:6.2f} (+{v[3]-v[2]:4.2f}) | all {v[4]:6.2f} (+{v[4]-v[3]:4.2f}) | staged+barrier {v[5]:6.2f} (+{v[5]-v[4]:4.2f})")
        prev = v[5]
