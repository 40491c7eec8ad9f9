"""psm_solve on registered buffers for meshes of growing size: the PCIe-reading stage kernel against the DMA copy (PSM_MESH_STAGE_MAX=0).
    python tools/attic/mesh_stage_crossover.py"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    from psm_amd import SolverModule, synthetic
    Lx, Ly = float(sys.argv[1]), float(sys.argv[2])
    model = synthetic.make_model("chapter5", p_in=32, p_out=32, seed_pca=4321, seed_w=11)
    array, top, obst = synthetic.channel_mesh(Lx=Lx, Ly=Ly, h=0.012, cx=1.0, R=0.15)
    sm = SolverModule(model, (1.0, 0.536133, 0.999023, 0.510742), geometry="native")
    sm.init_func(array, top, obst)
    cells, out = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
    sm.pin(cells, out)
    for _ in range(30): sm.py_func(cells, out=out)
    t0 = time.perf_counter()
    for _ in range(300): sm.py_func(cells, out=out)
    print(f"{array.shape[0]:7d} cells, grid {sm._sur.ny}x{sm._sur.nx}: {(time.perf_counter() - t0) / 300 * 1e6:7.1f} us per call")
else:
    for Lx, Ly in ((3.0, 1.2), (4.0, 1.6), (5.0, 2.0), (7.0, 2.0), (9.0, 2.0)):
        for mx in ("1000000000", "0"):
            env = dict(os.environ, PSM_MESH_STAGE_MAX=mx)
            out = subprocess.run([sys.executable, __file__, str(Lx), str(Ly)], env=env, capture_output=True, text=True).stdout.strip()
            print(("stage kernel: " if mx != "0" else "DMA copy    : ") + out, flush=True)
