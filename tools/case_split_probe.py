"""BASELINE configs[3] on one card: the same TOTAL case batch run as H handles x (total / H) cases, each handle on its own
HIP stream with its own bound geometries, one step = one call per handle, issued round-robin from one host thread.
    python tools/case_split_probe.py [total=64] [steps=600]
Prints us per step (all `total` cases) and solves/s for H = 1, 2, 4, 8 (those that divide the batch), and checks that the
split batches give the fields of the single call (max abs difference; the summation order of the M-tiled encode depends on
the batch size, so the last bits may differ)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import bench, psm_amd
from psm_amd import synthetic

total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
variant, NY, NX, _, prec, _ = bench.WORKLOADS["config3"]
model = synthetic.make_model(variant)
allc = synthetic.random_obstacle_cases(total, NY, NX, seed=3).astype(np.float32)
d_all = bench.to_device(torch, allc)
ref = None
for H in (1, 2, 4, 8, 16):
    if total % H or total // H < 1:
        continue
    n = total // H
    surs = [psm_amd.GridSurrogate(model, NY, NX, max_cases=n, precision=prec) for _ in range(H)]
    streams = [torch.cuda.Stream() for _ in range(H)]
    d_out = torch.empty((total, NY, NX, model.c_out), dtype=torch.float32, device="cuda")
    in_ptr = [d_all[k * n:(k + 1) * n].data_ptr() for k in range(H)]
    out_ptr = [d_out[k * n:(k + 1) * n].data_ptr() for k in range(H)]
    try:
        for k in range(H):
            assert surs[k].bind_geometry(in_ptr[k], on_device=True, n_cases=n)

        def step():
            for k in range(H):
                surs[k].solve_device(in_ptr[k], n, out_ptr[k], streams[k].cuda_stream)
        for _ in range(60):
            step()
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        got = d_out.cpu().numpy()
        if ref is None:
            ref = got
        diff = float(np.abs(got - ref).max())
        trips = sum(int(s.guard_trips) for s in surs)
        print(f"total {total:3d} cases as {H:2d} handle(s) x {n:2d}: {1e6 * best / steps:8.2f} us per step, {total * steps / best:10.0f} solves/s, "
              f"max |field - single call| {diff:.3e}, guard trips {trips}", flush=True)
    finally:
        for s in surs:
            s.close()
