"""Convolutional path: eager launches against one hipGraph replay per forward pass (captured with torch.cuda.graph on the
stream the library launches on).  usage: unet_graph_bench.py [n_cases] [precision]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
W = synthetic.unet_he_weights(seed=7)
g = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(n)])
with UNetSurrogate(W, 256, 256, max_cases=n, precision=prec) as net:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(n, 256, 256, 1))
    st = torch.cuda.Stream()
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
    eager = timeit(lambda: net.forward_device(d_in.ptr, n, d_out.ptr, st.cuda_stream))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        net.forward_device(d_in.ptr, n, d_out.ptr, st.cuda_stream)
    graph = timeit(lambda: gr.replay())
    print(f"UNet-S {prec} x{n}: eager {eager*1e6:7.1f} us/step, one graph replay per step {graph*1e6:7.1f} us/step")
