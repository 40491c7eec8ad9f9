"""Throughput of the other BASELINE configs (not the bench headline): solves/s with inputs resident in HBM."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import psm_amd
from psm_amd import synthetic

def run(name, model, ny, nx, n_cases, steps=1500, warm=150, precision="f32", bind=False):
    grids = synthetic.random_obstacle_cases(n_cases, ny, nx, seed=3).astype(np.float32)
    with psm_amd.GridSurrogate(model, ny, nx, max_cases=n_cases, precision=precision) as sur:
        d_in = torch.from_numpy(grids).pin_memory().cuda()
        if bind:
            assert sur.bind_geometry(d_in.data_ptr(), on_device=True, n_cases=n_cases)
            name += " [bound]"
        d_out = torch.empty((n_cases, ny, nx, model.c_out), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for i in range(warm): sur.solve_device(d_in.data_ptr(), n_cases, d_out.data_ptr(), st)
        dt = 1e9
        for rep in range(3):                      # best of 3: single runs show sporadic slow phases on a shared box
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps): sur.solve_device(d_in.data_ptr(), n_cases, d_out.data_ptr(), st)
            torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
        prof = sur.profile(d_in.data_ptr(), n_cases, d_out.data_ptr())
    print(f"{name:34s} B={sur.B:3d} cases/step={n_cases:3d}  {dt/steps*1e6:8.1f} us/step  {n_cases*steps/dt:10.0f} solves/s  "
          + " ".join(f"{k}={v*1e3:.1f}" for k, v in prof.items()))

run("config0 chapter5 128x128 P45/48", synthetic.make_model("chapter5", p_in=45, p_out=48), 128, 128, 1)
run("config1 gradp 256x256 b1", synthetic.make_model("gradp"), 256, 256, 1)
run("config2 deltas 256x256 b1", synthetic.make_model("deltas"), 256, 256, 1)
run("config3 deltas 256x256 x8", synthetic.make_model("deltas"), 256, 256, 8, steps=3000)
run("config3 deltas 256x256 x64", synthetic.make_model("deltas"), 256, 256, 64, steps=400, warm=20)
run("deltas 512x512 b1 (f32)", synthetic.make_model("deltas"), 512, 512, 1, steps=3000)
run("config4 deltas 512x512 b1 bf16", synthetic.make_model("deltas"), 512, 512, 1, steps=3000, precision="bf16")
run("config1 gradp 256x256 b1 bf16", synthetic.make_model("gradp"), 256, 256, 1, precision="bf16")
# geometry bound once per case slot (psm_bind_geometry_cases): 6 launches (7 for batches) instead of 8 (9)
run("config0 chapter5 128x128 P45/48", synthetic.make_model("chapter5", p_in=45, p_out=48), 128, 128, 1, bind=True)
run("config1 gradp 256x256 b1", synthetic.make_model("gradp"), 256, 256, 1, bind=True)
run("config2 deltas 256x256 b1", synthetic.make_model("deltas"), 256, 256, 1, bind=True)
run("config3 deltas 256x256 x8", synthetic.make_model("deltas"), 256, 256, 8, steps=3000, bind=True)
run("config3 deltas 256x256 x64", synthetic.make_model("deltas"), 256, 256, 64, steps=400, warm=20, bind=True)
run("deltas 512x512 b1 (f32)", synthetic.make_model("deltas"), 512, 512, 1, steps=3000, bind=True)
