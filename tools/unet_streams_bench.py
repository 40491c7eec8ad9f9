"""Convolutional path: the same 8 cases per step as ONE launch sequence of 8 cases, or as G handles of 8/G cases each on
their own streams (do the latency-bound deep layers of one group overlap the other group's?).
usage: unet_streams_bench.py [groups] [precision]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n = 8 // G
W = synthetic.unet_he_weights(seed=7)
g = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(n)])
nets = [UNetSurrogate(W, 256, 256, max_cases=n, precision=prec) for _ in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]
d_in = [DeviceArray(g) for _ in range(G)]
d_out = [DeviceArray(shape=(n, 256, 256, 1)) for _ in range(G)]
def step():
    for k in range(G):
        nets[k].forward_device(d_in[k].ptr, n, d_out[k].ptr, streams[k].cuda_stream)
for i in range(30): step()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    N = 200
    t0 = time.perf_counter()
    for i in range(N): step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / N)
print(f"UNet-S {prec} 8 cases per step as {G} x {n}: {best*1e6:8.1f} us/step  {8/best:9.0f} solves/s")
for x in nets: x.close()
