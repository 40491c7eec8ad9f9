import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import psm_oracle as orc
from psm_amd import synthetic, surrogate, GridSurrogate
ny=nx=256
for obst in ("circle","none"):
    grid = synthetic.channel_grid(ny,nx,seed=8,obstacle=obst).astype(np.float32)
    yy,xx=np.meshgrid(np.arange(ny),np.arange(nx),indexing="ij")
    truth=np.stack([np.sin(xx/50.0+c)+np.cos(yy/31.0) for c in range(2)],-1)
    lay=orc.block_layout("gradp",ny,nx)
    bp=orc.extract_blocks(truth,lay,2).astype(np.float32)
    fh,offs_h,sh_h=surrogate.debug_reassemble_host("gradp",grid,bp,2)
    model = synthetic.make_model("gradp", p_in=8, p_out=8)
    with GridSurrogate(model, ny, nx) as sur:
        fd = sur.reassemble(grid, bp)
        offs_d = sur.stage("offsets")[0]; sh_d = sur.stage("shift")[0]
    np.set_printoptions(precision=4,linewidth=220)
    print(obst, "max field diff", np.abs(fd-fh).max())
    print("offs host", offs_h[0]); print("offs dev ", offs_d[0])
    print("shift", sh_h, sh_d)
