import numpy as np, sys
sys.path.insert(0,'.')
import psm_amd, torch
from psm_amd import synthetic
model = synthetic.make_model("gradp")
grid = synthetic.channel_grid(256,256,seed=1).astype(np.float32)
with psm_amd.GridSurrogate(model,256,256) as sur:
    d_in = torch.from_numpy(grid).cuda(); d_out = torch.empty((256,256,2),dtype=torch.float32,device="cuda")
    for k in ("encode","decode","strips","paste","reduce","mlp"):
        for i in range(50): sur.solve_device(d_in.data_ptr(),1,d_out.data_ptr(),0)
        sur.enable_kernel_timing(k, True, 1)
        for i in range(500): sur.solve_device(d_in.data_ptr(),1,d_out.data_ptr(),0)
        t,n=sur.kernel_timing(k); sur.enable_kernel_timing(k,False)
        print(k, "event avg us", t/n*1e3)
    print("empty pair overhead us", sur.event_pair_overhead_ms(500)*1e3)
