"""Inside one convolution layer of UNet-S (diagnostic build): workgroup-0 stamps -- start, prologue done, then per chunk:
MFMA start, MFMA done, LDS stores done, barrier passed -- and the end of the epilogue."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from psm_amd import UNetSurrogate, synthetic
W = synthetic.unet_he_weights(seed=7)
import os
n = int(os.environ.get("N_CASES", "1")); prec = os.environ.get("PRECISION", "f32")        # N_CASES=8 PRECISION=bf16 python tools/attic/unet_stamps.py 7
g = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(n)])
names = ['enc0a','enc0b','enc1a','enc1b','enc2a','enc2b','enc3a','enc3b','enc4a','enc4b','dec3a','dec3b','dec2a','dec2b','dec1a','dec1b','dec0a','dec0b','head']
with UNetSurrogate(W, 256, 256, max_cases=n, precision=prec) as net:
    net.forward(g)
    for idx in [int(a) for a in sys.argv[1:]] or [1, 15]:
        acc = []
        for rep in range(20):
            st = np.zeros(64, np.float32)
            net._chk(net.lib.psm_unet_debug_run_layer(net.h, idx, st.ctypes.data_as(C.POINTER(C.c_float))))
            acc.append(st)
        st = np.median(np.array(acc), axis=0)
        print(f"{names[idx]}: start {st[0]:.2f}  prologue done {st[1]:.2f}")
        c = 0
        while 5 + 4 * c < 63 and st[2 + 4 * c] >= 0:
            print(f"   chunk {c}: mfma {st[2+4*c]:.2f} -> {st[3+4*c]:.2f} ({st[3+4*c]-st[2+4*c]:.2f})  stores done {st[4+4*c]:.2f}  barrier {st[5+4*c]:.2f}")
            c += 1
        print(f"   end {st[63]:.2f}")
