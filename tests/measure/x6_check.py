"""Accuracy and speed of the x6 arithmetic (PSM_X6 bit 0: encode, bit 1: decode) against the exact-float32 MFMA path:
rel-L2 of the scaled PCA coefficients and of the fields against the float64 oracle on BASELINE configs 1 and 3, and the
device-resident time per step.  Run once per PSM_X6 value (the mode is read when the library plans)."""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):        # 16-CPU quota on a 256-CPU box: keep the BLAS pool small
    os.environ.setdefault(_v, "16")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import psm_amd
from psm_amd import synthetic
from oracle import psm_oracle as orc
from bench import oracle_model

def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))

for variant, ny, nx, nc in (("gradp", 256, 256, 1), ("deltas", 256, 256, 8), ("deltas", 256, 256, 64)):
    model = synthetic.make_model(variant)
    grids = synthetic.random_obstacle_cases(nc, ny, nx, seed=3).astype(np.float32) if nc > 1 else synthetic.channel_grid(ny, nx, seed=1).astype(np.float32)[None]
    with psm_amd.GridSurrogate(model, ny, nx, max_cases=nc) as sur:
        f = sur.solve(grids)
        x = sur.stage("x_input", nc)[:sur.B]
        d_in, d_out = torch.from_numpy(grids).pin_memory().cuda(), torch.empty((nc, ny, nx, model.c_out), device="cuda")
        sur.bind_geometry(d_in.data_ptr(), on_device=True, n_cases=nc)
        for _ in range(50):
            sur.solve_device(d_in.data_ptr(), nc, d_out.data_ptr(), 0)
        sur.synchronize()
        n = 1000 if nc < 64 else 300
        t0 = time.perf_counter()
        for _ in range(n):
            sur.solve_device(d_in.data_ptr(), nc, d_out.data_ptr(), 0)
        sur.synchronize()
        dt = (time.perf_counter() - t0) / n
        fb = torch.empty(d_out.shape, dtype=d_out.dtype, pin_memory=True).copy_(d_out).numpy()
    sol = orc.solve_grid(grids[0].astype(np.float64), oracle_model(model))      # after the timing: no BLAS threads spinning under it
    print(f"PSM_X6={os.environ.get('PSM_X6', 'default')} {variant} x{nc}: x_input rel-L2 {rel(x, sol.x_input):.2e}  fields rel-L2 {rel(f[0], sol.fields):.2e} "
          f"(bound {rel(fb[0], sol.fields):.2e})  {dt * 1e6:.1f} us per step = {nc / dt:.0f} solves/s")
