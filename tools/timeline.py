"""Timeline of one steady-state solve from the diagnostic stamps build (make stamps;
PSM_LIB=.../libpsm_hip_stamps.so): s_memrealtime of workgroup 0 at the marked points of every
kernel, microseconds after the encode kernel's first stamp."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
NAMES = {0: "encode start", 1: "encode staged", 2: "encode end", 3: "encode tile 0 done", 4: "encode other tiles staged", 8: "rd1 start", 9: "rd1 slabs summed", 10: "rd1 L1 partials", 11: "rd1 end",
         16: "reduce start", 17: "reduce summed", 20: "decode start", 24: "decode requests issued", 25: "decode own rows staged (thread 0)", 21: "decode staged", 22: "decode mfma done", 23: "decode end",
         28: "strips start", 29: "strips loads landed", 30: "strips end", 36: "assemble start", 37: "assemble phase1", 38: "assemble pre",
         39: "assemble chain", 40: "assemble post", 41: "assemble end"}
for l in range(4):
    NAMES[44 + 4 * l] = f"dense{l} start"; NAMES[45 + 4 * l] = f"dense{l} mfma done"; NAMES[46 + 4 * l] = f"dense{l} end"
variant = sys.argv[1] if len(sys.argv) > 1 else "gradp"
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BIND = len(sys.argv) > 3 and sys.argv[3] == "bind"     # geometry-bound path: decode slots = decode+chain+paste, 39 = chain done
model = synthetic.make_model(variant)
grid = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(NC)])
with psm_amd.GridSurrogate(model, 256, 256, max_cases=NC) as sur:
    d_in, d_out = DeviceArray(grid), DeviceArray(shape=(NC, 256, 256, model.c_out))
    if BIND: assert sur.bind_geometry(d_in.ptr, on_device=True, n_cases=NC)
    acc = []
    for it in range(60):
        for k in range(20): sur.solve_device(d_in.ptr, NC, d_out.ptr, 0)
        sur.synchronize()
        out = np.zeros(64, np.float32)
        sur._chk(sur.lib.psm_read_stage(sur.h, 6, out.ctypes.data_as(C.POINTER(C.c_float)), 64))
        acc.append(out.copy())
    a = np.array(acc[10:])
    a = a - a[:, :1]
    med = np.median(a, axis=0)
    order = [k for k in np.argsort(med) if acc[-1][k] >= 0 and k in NAMES]
    prev = 0.0
    for k in order:
        print(f"{med[k]:7.2f}  (+{med[k]-prev:5.2f})  {NAMES[k]}")
        prev = med[k]
