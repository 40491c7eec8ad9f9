"""The reference's shipped solver case shape: 400 x 3000 grid, Chapter-5 layout (104 blocks), the real 45 / 48 component
network (python_module.py:103-134) -- solves/s with the input resident in HBM, general path and geometry bound."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import psm_amd
from psm_amd import synthetic
import cases
W, maxs, maxs_pca = cases.real_chapter5_weights()
model = synthetic.make_model("chapter5", p_in=45, p_out=48, weights=W)
model.in_a, model.out_a = float(maxs_pca[0]), float(maxs_pca[1])
grid = synthetic.channel_grid(400, 3000, seed=1).astype(np.float32)
for bind in (False, True):
    with psm_amd.GridSurrogate(model, 400, 3000) as sur:
        d_in = torch.from_numpy(grid[None]).pin_memory().cuda(); d_out = torch.empty((1, 400, 3000, 1), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if bind: assert sur.bind_geometry(d_in.data_ptr(), on_device=True)
        for i in range(200): sur.solve_device(d_in.data_ptr(), 1, d_out.data_ptr(), st)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(1000): sur.solve_device(d_in.data_ptr(), 1, d_out.data_ptr(), st)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 1000)
        profs = [sur.profile(d_in.data_ptr(), 1, d_out.data_ptr()) for _ in range(5)]
        prof = {k: min(p[k] for p in profs) for k in profs[0]}
        print('   groups (event-separated, us): ' + ' '.join(f'{k}={v*1e3:.1f}' for k, v in prof.items()))
        print(f"400x3000 chapter5, B={sur.B}, {'geometry bound' if bind else 'general path  '}: {best*1e6:7.1f} us per solve = {1/best:8.0f} solves/s")
