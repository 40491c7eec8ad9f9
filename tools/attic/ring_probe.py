"""Host-buffer entries of BASELINE config 1 through psm_bench_host (C++ loop inside the library):
    python tools/attic/ring_probe.py [mode depth steps] ...   (triples; default: every mode)
mode 0 sync pageable, 1 ring pageable, 2 ring registered, 3 ring zero-copy (include/psm.h: psm_bench_host)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
import psm_amd
from psm_amd import synthetic
model = synthetic.make_model("gradp")
grids = np.ascontiguousarray(np.stack([synthetic.channel_grid(256, 256, seed=1 + i).astype(np.float32)[None] for i in range(4)]))
for g in grids[1:]:
    g[..., 2] = grids[0][..., 2]
args = [int(a) for a in sys.argv[1:]]
runs = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(0, 1, 2000), (1, 3, 2000), (1, 6, 2000), (2, 1, 2000), (2, 2, 2000), (2, 3, 2000), (2, 4, 2000), (2, 6, 2000), (2, 8, 2000), (3, 1, 2000), (3, 2, 2000), (3, 3, 2000), (3, 4, 2000), (3, 6, 2000), (3, 8, 2000)]
with psm_amd.GridSurrogate(model, 256, 256) as sur:
    assert sur.bind_geometry(grids[0, 0])
    for mode, depth, steps in runs:
        sec = C.c_double()
        sur._chk(sur.lib.psm_bench_host(sur.h, grids.ctypes.data_as(C.POINTER(C.c_float)), 4, 1, mode, depth, steps, 100, C.byref(sec), None))
        print(f"mode {mode} depth {depth}: {steps / sec.value:9.0f} solves/s ({1e6 * sec.value / steps:6.1f} us per solve)", flush=True)
