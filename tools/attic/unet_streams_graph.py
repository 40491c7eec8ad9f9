"""Conv path: n cases as S independent chains of n/S cases on S streams, captured as ONE hipGraph with S parallel
branches (fork / join through events) and replayed once per step -- against one handle x n cases launched plainly.
The plain-launch form of the same split (tools/attic/unet_streams.py) is bound by the host's launch rate; a graph replay is not.
usage: unet_streams_graph.py [n_cases] [precision] [size]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
W = synthetic.unet_he_weights(seed=7)


def timeit(fn, N=300):
    for i in range(30): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(N): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    return best


for S in (1, 2, 4):
    m = n // S
    if m < 1: break
    g = np.stack([synthetic.channel_grid(size, size, seed=1 + k).astype(np.float32) for k in range(m)])
    nets = [UNetSurrogate(W, size, size, max_cases=m, precision=prec, autotune=(len(sys.argv) > 4)) for _ in range(S)]
    d_in = [DeviceArray(g) for _ in range(S)]; d_out = [DeviceArray(shape=(m, size, size, 1)) for _ in range(S)]
    sts = [torch.cuda.Stream() for _ in range(S)]

    def chains():
        for s in range(1, S): sts[s].wait_stream(sts[0])
        for s in range(S): nets[s].forward_device(d_in[s].ptr, m, d_out[s].ptr, sts[s].cuda_stream)
        for s in range(1, S): sts[0].wait_stream(sts[s])

    eager = timeit(chains)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=sts[0]):
        chains()
    graph = timeit(lambda: gr.replay())
    fl = nets[0].flops * n
    print(f"UNet-S {prec} {size}x{size} {n} cases as {S} chain(s) x {m}: plain launches {eager*1e6:7.1f} us/step, one graph replay "
          f"{graph*1e6:7.1f} us/step = {n/graph:8.0f} solves/s {fl/graph/1e12:6.1f} TFLOP/s", flush=True)
    del gr
    for x in nets: x.close()
