"""One drawn channel mesh through SolverModule with geometry='native' and 'scipy' and through the NumPy oracle: where do the pressures differ?
    python tests/measure/mesh_native_vs_scipy.py Lx Ly h R cx cy seed p_in p_out"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from psm_amd import SolverModule, synthetic
from oracle import psm_oracle as orc
from bench import oracle_model
Lx, Ly, h, R, cx, cy = [float(v) for v in sys.argv[1:7]]
mseed, p_in, p_out = [int(v) for v in sys.argv[7:10]]
model = synthetic.make_model("chapter5", p_in=p_in, p_out=p_out, seed_pca=1, seed_w=2)
array, top, obst = synthetic.channel_mesh(Lx=Lx, Ly=Ly, h=h, seed=mseed, cx=cx, cy=cy, R=R, step=0)
geo = orc.init_geometry(array, top, obst)
ref = orc.py_func_mesh(array, geo, oracle_model(model), cases.MESH_MAXS)[0]
out = {}
for g in ("scipy", "native"):
    sm = SolverModule(model, cases.MESH_MAXS, geometry=g)
    sm.init_func(array, top, obst, 0)
    out[g] = sm.py_func(array, 0)
    grid = sm._sur.stage("x_input")
    d = np.abs(out[g] - ref)
    bad = d > 2e-4 * np.abs(ref).max()
    print(f"{g:7s}: max rel err {d.max() / np.abs(ref).max():.3e}, cells off by > 2e-4: {int(bad.sum())} of {len(ref)}; x_input |max| {np.abs(grid).max():.4f}")
    if bad.any():
        idx = np.argsort(-d)[:8]
        for i in idx:
            print(f"    cell {i}: C=({array[i,2]:.4f}, {array[i,3]:.4f}) p={out[g][i]:.5f} ref={ref[i]:.5f} p_prev={array[i,4]:.5f}")
print("native vs scipy: max abs diff", np.abs(out["native"] - out["scipy"]).max(), "of", np.abs(ref).max())
