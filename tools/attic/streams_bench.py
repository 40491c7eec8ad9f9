"""Independent case streams sharing one GPU: S handles on S HIP streams, solves issued round-robin."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import psm_amd
from psm_amd import synthetic
model = synthetic.make_model("gradp")
BIND = len(sys.argv) > 1 and sys.argv[1] == "bind"     # one geometry per stream, bound once (6 launches per solve)
for S in (1, 2, 3, 4, 6, 8):
    surs = [psm_amd.GridSurrogate(model, 256, 256) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    d_in = [torch.from_numpy(synthetic.channel_grid(256, 256, seed=1 + i).astype(np.float32)).pin_memory().cuda() for i in range(S)]
    d_out = [torch.empty((256, 256, 2), dtype=torch.float32, device="cuda") for _ in range(S)]
    if BIND:
        for k in range(S): assert surs[k].bind_geometry(d_in[k].data_ptr(), on_device=True)
    def step(i):
        k = i % S
        surs[k].solve_device(d_in[k].data_ptr(), 1, d_out[k].data_ptr(), streams[k].cuda_stream)
    for i in range(300): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 4000
    for i in range(N): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'bound ' if BIND else ''}streams={S}: {N/dt:9.0f} solves/s  ({dt/N*1e6:.1f} us per solve)")
    for s in surs: s.close()
