import numpy as np, sys, ctypes as C
sys.path.insert(0,'.')
import psm_amd
from psm_amd import synthetic, _lib
import torch
model = synthetic.make_model("gradp")
grid = synthetic.channel_grid(256,256,seed=1).astype(np.float32)
with psm_amd.GridSurrogate(model,256,256) as sur:
    d_in = torch.from_numpy(grid).cuda(); d_out = torch.empty((256,256,2),dtype=torch.float32,device="cuda")
    acc=[]
    for it in range(60):
        for k in range(3): sur.solve_device(d_in.data_ptr(),1,d_out.data_ptr(),0)
        sur.synchronize()
        out=np.zeros(64,np.float32)
        sur._chk(sur.lib.psm_read_stage(sur.h,5,out.ctypes.data_as(C.POINTER(C.c_float)),64))
        acc.append(out.copy())
    a=np.median(np.array(acc[10:]),axis=0)
    print("encode  : stage(A loads,B issue,LDS,barrier)=%.2f  mfma+store=%.2f"%(a[0],a[1]))
    print("reduce  : loads+sum=%.2f"%a[8])
    print("dense1  : loads+mfma=%.2f  reduce+store=%.2f"%(a[48],a[49]))
    print("decode  : stage=%.2f  mfma=%.2f  store=%.2f"%(a[20],a[21],a[22]))
    print("strips  : loads=%.2f  slots=%.2f"%(a[28],a[29]))
    print("assemble: phase1=%.2f  pre=%.2f  chain=%.2f  post=%.2f  paste=%.2f"%tuple(a[36:41]))
