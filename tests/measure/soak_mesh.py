"""Randomised soak of the solver boundary (cells[N,5] float64 -> p[N] float64: init_func + py_func, PythonComm.H contract)
against the NumPy oracle's init_geometry + py_func on drawn channel meshes: channel size, cell size (so the 5 mm grid is 1-4
block rows x 2-7 block columns), jitter seed, obstacle position and radius, component counts; two time steps per case.

    python tests/measure/soak_mesh.py [trials] [seed]
"""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from psm_amd import SolverModule, synthetic
from oracle import psm_oracle as orc
from bench import oracle_model

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(seed)
worst, n_fb, t0 = 0.0, 0, time.time()
for trial in range(trials):
    Lx, Ly = float(rng.uniform(1.0, 3.2)), float(rng.uniform(0.66, 1.9))
    h = float(rng.uniform(0.006, 0.012))
    R = float(rng.uniform(0.04, min(0.2, Ly / 4)))
    cx, cy = float(rng.uniform(0.2, Lx - 0.3)), float(rng.uniform(-Ly / 2 + R + 0.05, Ly / 2 - R - 0.05))
    mseed = int(rng.integers(1 << 20))
    p_in, p_out = int(rng.integers(2, 100)), int(rng.integers(2, 100))
    model = synthetic.make_model("chapter5", p_in=p_in, p_out=p_out, seed_pca=int(rng.integers(1 << 20)), seed_w=int(rng.integers(1 << 20)))
    info = dict(trial=trial, Lx=round(Lx, 3), Ly=round(Ly, 3), h=round(h, 4), R=round(R, 3), cx=round(cx, 3), cy=round(cy, 3), seed=mseed, p_in=p_in, p_out=p_out)
    array, top, obst = synthetic.channel_mesh(Lx=Lx, Ly=Ly, h=h, seed=mseed, cx=cx, cy=cy, R=R, step=0)
    geo = orc.init_geometry(array, top, obst)
    om = oracle_model(model)
    sm = SolverModule(model, cases.MESH_MAXS)
    if sm.init_func(array, top, obst, 0) != 0:
        raise SystemExit(f"init_func failed: {info}")
    for step in (0, 2):
        a = synthetic.channel_mesh(Lx=Lx, Ly=Ly, h=h, seed=mseed, cx=cx, cy=cy, R=R, step=step)[0]
        p = sm.py_func(a, 0)
        ref = orc.py_func_mesh(a, geo, om, cases.MESH_MAXS)[0]
        fb = ref == a[:, 4]                                             # near-wall / NaN fallback: previous pressure kept, bit for bit
        n_fb += int(fb.sum())
        if not np.array_equal(p[fb], a[fb, 4]):
            raise SystemExit(f"fallback cells differ: {info} step {step}")
        if np.isnan(p).any() != np.isnan(ref).any():
            raise SystemExit(f"NaN presence differs: {info} step {step}")
        err = float(np.nanmax(np.abs(p - ref)) / max(np.nanmax(np.abs(ref)), 1e-9))
        worst = max(worst, err)
        if err > 2e-4:
            raise SystemExit(f"mismatch {err:.2e}: {info} step {step} grid {geo.ny}x{geo.nx} cells {len(a)}")
    if trial % 5 == 4:
        print(f"trial {trial + 1}/{trials}: last grid {geo.ny}x{geo.nx}, {len(array)} cells; worst rel err {worst:.2e}, {n_fb} fallback cells so far, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK OK: {trials} meshes x 2 steps, worst rel err {worst:.2e} (tolerance 2e-4), {n_fb} fallback cells identical, seed {seed}")
