"""Conv path: S handles (one stream each) x n/S cases launched back to back vs one handle x n cases."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
W = synthetic.unet_he_weights(seed=7)
for S in (1, 2, 4):
    m = n // S
    g = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(m)])
    nets = [UNetSurrogate(W, 256, 256, max_cases=m, precision=prec) for _ in range(S)]
    d_in = [DeviceArray(g) for _ in range(S)]; d_out = [DeviceArray(shape=(m, 256, 256, 1)) for _ in range(S)]
    def step():
        for s in range(S): nets[s].forward_device(d_in[s].ptr, m, d_out[s].ptr, 0)
    for i in range(30): step()
    for s in range(S): nets[s].synchronize()
    best = 1e9
    for rep in range(3):
        N = 200
        t0 = time.perf_counter()
        for i in range(N): step()
        for s in range(S): nets[s].synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    print(f"{prec} {n} cases as {S} stream(s) x {m}: {best*1e6:8.1f} us/step  {n/best:9.0f} solves/s  {nets[0].flops*n/best/1e12:6.1f} TFLOP/s", flush=True)
    for x in nets: x.close()
