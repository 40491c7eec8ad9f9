"""Throughput of the convolutional path (UNet-S, 256x256x3 -> 256x256x1, float32): solves/s with the input
resident in HBM, and achieved FLOP/s against the dense f32 MFMA peak (157 TFLOP/s)."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
ny = nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
W = synthetic.unet_he_weights(seed=7)
g = np.stack([synthetic.channel_grid(ny, nx, seed=1 + k).astype(np.float32) for k in range(n)])
with UNetSurrogate(W, ny, nx, max_cases=n, precision=prec) as net:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(n, ny, nx, 1))
    for i in range(50): net.forward_device(d_in.ptr, n, d_out.ptr, 0)
    net.synchronize()
    best = 1e9
    for rep in range(3):
        N = 300
        t0 = time.perf_counter()
        for i in range(N): net.forward_device(d_in.ptr, n, d_out.ptr, 0)
        net.synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    fl = net.flops * n
    print(f"UNet-S {prec} {ny}x{nx} x{n}: {best*1e6:8.1f} us/step  {n/best:9.0f} solves/s  {fl/best/1e12:6.2f} TFLOP/s ({fl/best/157e12*100:.1f}% of f32 MFMA peak)")
    ms = np.min([net.profile(d_in.ptr, n, d_out.ptr)[0] for _ in range(5)], axis=0)
    wg = net.profile(d_in.ptr, n, d_out.ptr)[1]
    names = ['enc0a','enc0b','enc1a','enc1b','enc2a','enc2b','enc3a','enc3b','enc4a','enc4b','dec3a','dec3b','dec2a','dec2b','dec1a','dec1b','dec0a','dec0b','head']
    for i, ((k, ci, co), t, w) in enumerate(zip(synthetic.unet_conv_shapes(), ms, wg)):
        lvl = net._level(i)
        f = 2 * (ny >> lvl) * (nx >> lvl) * k ** 2 * ci * co * n
        print(f"  {names[i]:6s} {ci:4d}->{co:4d} @{ny >> lvl:4d}  wgs={w:5d}  {t*1e3:7.1f} us  {f/(t*1e-3)/1e12:6.1f} TFLOP/s")
