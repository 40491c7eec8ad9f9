"""Inside a fused level pair (diagnostic build, PSM_LIB=.../libpsm_hip_stamps.so): stamps of one persistent workgroup (its tile loop), microseconds after the earliest stamp.  usage: pair_stamps.py [n_cases] [layer ...]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from psm_amd import UNetSurrogate, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = synthetic.unet_he_weights(seed=7)
g = np.stack([synthetic.channel_grid(256, 256, seed=1 + k).astype(np.float32) for k in range(n)])
with UNetSurrogate(W, 256, 256, max_cases=n, precision="bf16") as net:
    net.forward(g)
    for idx in [int(a) for a in sys.argv[2:]] or [16]:
        acc = []
        for rep in range(10):
            st = np.zeros(64, np.float32)
            net._chk(net.lib.psm_unet_debug_run_layer(net.h, idx, st.ctypes.data_as(C.POINTER(C.c_float))))
            acc.append(st)
        st = np.median(np.array(acc), axis=0)
        if len(sys.argv) > 2 and idx in (2, 14):       # 32-channel pair: raw stamp list of the first tile
            print(f"layer {idx}: kernel entry {st[63]:6.2f};", " ".join(f"{v:6.2f}" for v in st[:24] if v >= 0))
            continue
        print(f"layer {idx}: kernel entry {st[63]:6.2f} (us after the earliest stamp); prologue stamps " + " ".join(f"{v:5.2f}" for v in st[56:63] if v >= 0))
        for it in range(7):
            if st[9 * it] < 0:
                break
            print(f"layer {idx} tile {it}: loop top {st[9*it]:6.2f} | written {st[9*it+1]:6.2f} barrier {st[9*it+2]:6.2f} | conv A1 (stem: next issued) {st[9*it+3]:6.2f} A2 (stem: conv A) {st[9*it+4]:6.2f} "
                  f"mid {st[9*it+5]:6.2f} barrier {st[9*it+6]:6.2f} | conv B {st[9*it+7]:6.2f} out {st[9*it+8]:6.2f}")
