"""densePCA_attention ('MLP_attention', 3 x 512 + attention block) on the 256 x 256 deltas shape: launches and time per solve, bound path.
    python tools/attic/attention_bench.py            (PSM_LN_FUSE=0: every LayerNormalization as its own launch)"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, psm_amd
from psm_amd import synthetic
from hipmem import DeviceArray
model = synthetic.make_model("deltas", arch="MLP_attention")
g = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)[None]
with psm_amd.GridSurrogate(model, 256, 256) as sur:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(1, 256, 256, 1))
    assert sur.bind_geometry(d_in.ptr, on_device=True, n_cases=1)
    kt = bench.time_kernels(sur, d_in.ptr, 1, d_out.ptr, 1000)
    for nm, us, n in kt:
        print(f"{nm:52s} {us:7.2f} us x {n / 1000:.0f}")
    for i in range(300): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize(); t0 = time.perf_counter()
    for i in range(2000): sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
    sur.synchronize(); dt = time.perf_counter() - t0
    print(f"back-to-back: {1e6 * dt / 2000:.2f} us per solve, {2000 / dt:.0f} solves/s")
