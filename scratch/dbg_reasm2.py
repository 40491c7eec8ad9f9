import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import psm_oracle as orc
from psm_amd import synthetic, surrogate, GridSurrogate
ny=nx=256
rng = np.random.default_rng(0)
grid = synthetic.channel_grid(ny,nx,seed=8).astype(np.float32)
open_grid = synthetic.channel_grid(ny,nx,seed=8,obstacle="none").astype(np.float32)
yy,xx=np.meshgrid(np.arange(ny),np.arange(nx),indexing="ij")
truth=np.stack([np.sin(xx/50.0+c)+np.cos(yy/31.0) for c in range(2)],-1)
lay=orc.block_layout("gradp",ny,nx)
bp=orc.extract_blocks(truth,lay,2)
model = synthetic.make_model("gradp", p_in=8, p_out=8)
fh,_,_=surrogate.debug_reassemble_host("gradp",open_grid,bp.astype(np.float32),2)
with GridSurrogate(model, ny, nx) as sur:
    f0 = sur.reassemble(grid, bp)
    f1 = sur.reassemble(grid, bp + rng.standard_normal((lay.B, 1, 1, 2)))
    f2 = sur.reassemble(open_grid, bp)
    f3 = sur.reassemble(open_grid, bp)
print("f0-f1", np.abs(f0-f1).max(), "f2-host", np.abs(f2-fh).max(), "f3-host", np.abs(f3-fh).max())
d = f2[...,0]-truth[...,0]; print(d.max()-d.min())
print(open_grid.flags, open_grid[...,:3].flags['C_CONTIGUOUS'])
