"""Per-launch dispatch times of the convolutional path with the planner's choices, for one (size, cases, precision):
    python tools/unet_layers.py 512 1 bf16 [autotune]
prints the plan (tile rows, channel tiles, split, role) and the dispatch-stamped time of every launch.  PSM_UNET_FORCE etc. apply."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from psm_amd import UNetSurrogate, synthetic
from hipmem import DeviceArray
ny = nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
tune = len(sys.argv) > 4 and sys.argv[4] == "autotune"
names = ['enc0a','enc0b','enc1a','enc1b','enc2a','enc2b','enc3a','enc3b','enc4a','enc4b','dec3a','dec3b','dec2a','dec2b','dec1a','dec1b','dec0a','dec0b','head']
W = synthetic.unet_he_weights(seed=7)
g = np.stack([synthetic.channel_grid(ny, nx, seed=1 + k).astype(np.float32) for k in range(n)])
with UNetSurrogate(W, ny, nx, max_cases=n, precision=prec, autotune=tune) as net:
    d_in, d_out = DeviceArray(g), DeviceArray(shape=(n, ny, nx, 1))
    for i in range(20): net.forward_device(d_in.ptr, n, d_out.ptr, 0)
    net.synchronize()
    tot = 0.0
    for first, convs, kname, us in net.time_kernels(d_in.ptr, n, d_out.ptr, steps=40):
        fl = sum(net.conv_flops(c) for c in convs) * n
        tot += us
        print(f"  {'+'.join(names[c] for c in convs):22s} plan={net.plan_info(first)}  {us:7.2f} us  {fl/us/1e6:7.1f} TFLOP/s  {kname[:70]}")
    print(f"UNet-S {prec} {ny}x{nx} x{n}: sum of launches {tot:.1f} us" + (f"  autotuned {net.autotuned['us_before']:.1f} -> {net.autotuned['us_after']:.1f}" if tune else ""))
