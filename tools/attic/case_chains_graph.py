"""Surrogate (PCA + MLP) path, BASELINE configs[3]: n cases per step as S independent chains of n/S cases (one handle and one
stream each, geometry bound per case slot), captured as ONE hipGraph with S parallel branches and replayed once per step --
against one handle x n cases launched plainly.  usage: case_chains_graph.py [n_cases]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import psm_amd
from psm_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = synthetic.make_model("deltas")
grids = synthetic.random_obstacle_cases(n, 256, 256, seed=3).astype(np.float32)


def timeit(fn, N=1500):
    for i in range(100): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(N): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    return best


for S in (1, 2, 4, 8):
    m = n // S
    if m < 1: break
    surs = [psm_amd.GridSurrogate(model, 256, 256, max_cases=m) for _ in range(S)]
    sts = [torch.cuda.Stream() for _ in range(S)]
    d_in = [torch.from_numpy(grids[s * m:(s + 1) * m].copy()).cuda() for s in range(S)]
    d_out = [torch.empty((m, 256, 256, model.c_out), dtype=torch.float32, device="cuda") for _ in range(S)]
    for s in range(S):
        assert surs[s].bind_geometry(d_in[s].data_ptr(), on_device=True, n_cases=m)

    def chains():
        for s in range(1, S): sts[s].wait_stream(sts[0])
        for s in range(S): surs[s].solve_device(d_in[s].data_ptr(), m, d_out[s].data_ptr(), sts[s].cuda_stream)
        for s in range(1, S): sts[0].wait_stream(sts[s])

    eager = timeit(chains)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=sts[0]):
        chains()
    graph = timeit(lambda: gr.replay())
    print(f"deltas 256x256 {n} cases as {S} chain(s) x {m}: plain launches {eager*1e6:7.1f} us/step, one graph replay "
          f"{graph*1e6:7.1f} us/step = {n/graph:8.0f} solves/s", flush=True)
    del gr
    for x in surs: x.close()
