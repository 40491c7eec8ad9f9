"""Seeded test cases shared by tests/golden/make_golden.py (which runs the
reference's statements on them) and the parity tests (which rebuild the same
inputs and compare the oracle / the HIP path with the stored outputs)."""
from __future__ import annotations

import os

import numpy as np

from psm_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> spec.  'band' = (y0, y1, x0, x1) rectangle zeroed in every channel (a
# large solid body) placed so that one block's bottom overlap strip holds no
# flow cell: drives the np.isnan(BC_ups[..]) branches without poisoning the field.
GOLDEN_CASES = {
    "gradp_272x288":      dict(variant="gradp", Ny=272, Nx=288, seed=11, p=32),
    "gradp_300x300":      dict(variant="gradp", Ny=300, Nx=300, seed=12, p=32),
    "gradp_nan_280x320":  dict(variant="gradp", Ny=280, Nx=320, seed=13, p=24, band=(64, 160, 64, 192), obstacle="none"),
    "deltas_256x256":     dict(variant="deltas", Ny=256, Nx=256, seed=21, p=32, scaler="std", U_max_norm=1.3, max_abs_p=0.51),
    "deltas_300x420":     dict(variant="deltas", Ny=300, Nx=420, seed=22, p=32, scaler="min_max", U_max_norm=0.8, max_abs_p=0.9),
    "deltas_nan_256x256": dict(variant="deltas", Ny=256, Nx=256, seed=23, p=24, scaler="max_abs", band=(96, 128, 32, 160), obstacle="none"),
    "chapter5_128x128_real": dict(variant="chapter5", Ny=128, Nx=128, real_weights=True),
    "chapter5_300x400":   dict(variant="chapter5", Ny=300, Nx=400, seed=31, p=32),
    "chapter5_nan_300x400": dict(variant="chapter5", Ny=300, Nx=400, seed=32, p=24, band=(116, 128, 156, 284), obstacle="none"),
    # Thesis_Work/Chapter4 evaluators run with their own trained models (BASELINE configs[0] as written):
    # `avance = int(0.75 * 128)` = 96 (M_u/Evaluation/Eval_dual_Dense_onlycil.py:499), stride 32
    "chapter4_Mu_128x128": dict(variant="chapter5", Ny=128, Nx=128, chapter4="M_u"),                 # cavity-like vortex, 4 blocks
    "chapter4_Mu_232x312": dict(variant="chapter5", Ny=232, Nx=312, seed=61, chapter4="M_u"),        # channel + cylinder, 35 blocks
    "chapter4_MfU_200x280": dict(variant="chapter5", Ny=200, Nx=280, seed=62, chapter4="M_fU"),      # f(U), SDF -> p: 2 input channels
    # the reference's conv1D_PCA network (NNs.py:75-124, architecture 'conv1D' of utils.py:452-454) as the evaluator's model:
    # built by the reference's own function (tests/golden/make_golden.py), weights stored as data in the fixture
    "deltas_conv1d_256x256": dict(variant="deltas", Ny=256, Nx=256, seed=81, p=32, scaler="std", conv1d=True),
    # the reference's densePCA_attention network (NNs.py:40-72, architecture 'MLP_attention' of utils.py:455-457), likewise built by
    # the reference's own function: which layers exist, their order, the residuals and the attention call are the reference's
    "deltas_attention_256x256": dict(variant="deltas", Ny=256, Nx=256, seed=82, p=32, scaler="std", attention=True),
}


# Grids whose last block row duplicates the previous one (p_i == 0): the reference does not define an answer there.
# U_to_gradP takes the mean of an empty slice (Eval_dual_Dense_onlycil.py:340) and the NaN offset reaches every cell
# through the global shift (:359); deltaU_to_deltaP raises a broadcast error (SM_call.py:335).  BASELINE configs[1]
# (256 rows, gradP) and configs[4] (512 rows, deltas) are such grids; `strict_degenerate=1` must reproduce exactly this.
DEGENERATE_CASES = {
    "gradp_degenerate_256x256":  dict(variant="gradp", Ny=256, Nx=256, seed=71, p=16),
    "deltas_degenerate_512x512": dict(variant="deltas", Ny=512, Nx=512, seed=72, p=16, scaler="std"),
}


# compute_in_block_error (utils.py:210-243): golden values in tests/golden/block_error_deltas.npz, labels = channel 3 of the grids
BLOCK_ERROR_CASES = ["deltas_256x256", "deltas_300x420", "deltas_nan_256x256"]

DENSE_GOLDEN_CASES = [k for k, v in GOLDEN_CASES.items() if not v.get("conv1d") and not v.get("attention")]     # Dense-stack networks (C port, bound path)


def real_chapter5_weights():
    d = np.load(os.path.join(GOLDEN_DIR, "chapter5_weights.npz"))
    n = len([k for k in d.files if k.startswith("W")])
    return [(d[f"W{i}"], d[f"b{i}"]) for i in range(n)], d["maxs"], d["maxs_PCA"]


def real_chapter4_weights(which: str):
    """Trained Chapter-4 networks of the reference as data (tests/golden/chapter4_weights.npz, written by
    make_golden.py with the build's own HDF5 reader): 'M_u' = M_u/trained_models/cil.h5 (32 -> 3x512 -> 32),
    'M_fU' = M_fU/Evaluation/model_first_.h5 (116 -> 3x512 -> 39) with that directory's maxs / maxs_PCA files."""
    d = np.load(os.path.join(GOLDEN_DIR, "chapter4_weights.npz"))
    n = len([k for k in d.files if k.startswith(which + "_W")])
    return [(d[f"{which}_W{i}"], d[f"{which}_b{i}"]) for i in range(n)], d["MfU_maxs"], d["MfU_maxs_PCA"]


def build(name: str):
    """-> (grid[Ny,Nx,C] float64, SurrogateModel) for a GOLDEN_CASES entry."""
    sp = GOLDEN_CASES[name] if name in GOLDEN_CASES else DEGENERATE_CASES[name]
    v, Ny, Nx = sp["variant"], sp["Ny"], sp["Nx"]
    if sp.get("chapter4"):
        which = sp["chapter4"]
        W, maxs, maxs_pca = real_chapter4_weights(which)
        c_in = 3 if which == "M_u" else 2
        model = synthetic.make_model("chapter5", p_in=W[0][0].shape[0], p_out=W[-1][0].shape[1], weights=W, c_in=c_in,
                                     seed_pca=4000 + c_in)
        model.ov, model.sdf_ch = 96, c_in - 1                # avance = int(0.75 * shape)
        if "seed" in sp:
            g = synthetic.channel_grid(Ny, Nx, seed=sp["seed"], extra_channels=1, cx=0.35, r=0.16)
        else:
            g = np.concatenate([synthetic.cavity_grid(Ny), np.zeros((Ny, Ny, 1))], axis=-1)
            g[..., 3] = np.cos(2 * np.pi * np.arange(Ny) / Ny)[None, :] * np.sin(np.pi * np.arange(Ny) / Ny)[:, None]
        if which == "M_fU":                                  # (f(U), SDF, p): 3 channels, 2 of them inputs
            fu = g[..., 0] ** 2 - 0.5 * g[..., 1]
            fu[g[..., 2] == 0] = 0.0
            g = np.stack([fu / np.abs(fu).max(), g[..., 2], g[..., 3]], axis=-1)
            model.in_a, model.out_a = 0.6, 12.0              # that directory's maxs_PCA holds a copy of `maxs`: synthetic scalers
        return g, model
    if sp.get("real_weights"):
        # BASELINE config 0: real trained MLP (45->512x3->48) of the reference's
        # Chapter-5 test case, its maxs_PCA scalers, synthetic PCA seed 1234.
        W, maxs, maxs_pca = real_chapter5_weights()
        model = synthetic.make_model("chapter5", p_in=W[0][0].shape[0], p_out=W[-1][0].shape[1], weights=W)
        model.in_a, model.out_a = float(maxs_pca[0]), float(maxs_pca[1])
        grid = synthetic.cavity_grid(Ny)
        grid[..., 2] *= 0.5           # PM:292 keeps the SDF un-normalised: any non-unit scale
        return grid, model
    extra = {"gradp": 3, "deltas": 2, "chapter5": 0}[v]
    grid = synthetic.channel_grid(Ny, Nx, seed=sp["seed"], extra_channels=extra,
                                  obstacle=sp.get("obstacle", "circle"))
    if "band" in sp:
        y0, y1, x0, x1 = sp["band"]
        grid[y0:y1, x0:x1, :] = 0.0
    model = synthetic.make_model(v, p_in=sp["p"], p_out=sp["p"], scaler_kind=sp.get("scaler"),
                                 seed_pca=1000 + sp["seed"], seed_w=sp["seed"])
    if sp.get("conv1d"):          # weights drawn by the reference-built network in make_golden.py, kept in the fixture
        f = os.path.join(GOLDEN_DIR, f"{name}.npz")
        if os.path.exists(f):
            d = np.load(f)
            nc = len([k for k in d.files if k.startswith("convK")])
            model.conv1d = [(d[f"convK{i}"], d[f"convb{i}"]) for i in range(nc)]
            model.weights = [(d["denseW"], d["denseb"])]
    if sp.get("attention"):       # likewise: parameters drawn by the reference-built densePCA_attention network
        f = os.path.join(GOLDEN_DIR, f"{name}.npz")
        if os.path.exists(f):
            d = np.load(f)
            nd = len([k for k in d.files if k.startswith("denseW")])
            model.weights = [(d[f"denseW{i}"], d[f"denseb{i}"]) for i in range(nd)]
            nl = len([k for k in d.files if k.startswith("ln_gamma")])
            att = {k: d["att_" + k] for k in ("Wv", "bv", "Wo", "bo")}
            # the fixture does not carry the query / key projections: over the reference's sequence of length 1 the softmax is 1
            # whatever they are (tests/test_attention.py proves it on the oracle) -- any values of the right shape do
            rq = np.random.default_rng(4242)
            for k in ("q", "k"):
                att["W" + k] = (rq.standard_normal(att["Wv"].shape) * 0.05).astype(np.float32)
                att["b" + k] = (rq.standard_normal(att["bv"].shape) * 0.05).astype(np.float32)
            model.attention = dict(att, ln=[(d[f"ln_gamma{i}"], d[f"ln_beta{i}"]) for i in range(nl)], eps=float(d["ln_eps"]))
    if v == "deltas":
        model.out_scale = sp.get("max_abs_p", 1.0) * sp.get("U_max_norm", 1.0) ** 2
    return grid, model


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))


MESH_MAXS = (1.0, 0.536133, 0.999023, 0.510742)      # the reference test case's `maxs` file


def build_mesh_case(step: int = 0):
    """Solver-side case: (array[N,5], top, obst, model, maxs) -- 140x300 grid, 8 chapter5 blocks."""
    array, top, obst = synthetic.channel_mesh(step=step)
    model = synthetic.make_model("chapter5", p_in=32, p_out=32, seed_pca=4321, seed_w=11)
    return array, top, obst, model, MESH_MAXS


def build_filter_case():
    """Inputs of the optional post-steps of assemble_prediction (SM_call.py:352-363): decoded blocks of
    the deltas_256x256 case (oracle), a deltaU-change image and a previous delta-p image."""
    from oracle import psm_oracle as orc
    grid, model = build("deltas_256x256")
    sc = orc.Scaler(model.scaler_kind, model.in_a, model.in_b, model.out_a, model.out_b)
    om = orc.Model(model.variant, model.c_in, model.c_out, model.comp_in, model.mean_in, model.comp_out, model.mean_out,
                   model.weights, sc, model.out_scale, model.S, model.ov, model.sdf_ch)
    sol = orc.solve_grid(grid, om)
    rng = np.random.default_rng(77)
    ny, nx = grid.shape[:2]
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    dU = np.abs(np.sin(xx / 40.0) * np.cos(yy / 25.0)) + 0.05 * rng.random((ny, nx))
    dU /= dU.max()
    dPprev = 0.3 * np.cos(xx / 33.0 + 1.0) * np.sin(yy / 47.0) + 0.02 * rng.standard_normal((ny, nx))
    return grid, model, sol.block_pred[..., 0], dU, dPprev


def build_integration_case():
    """U_to_gradP integration (Eval_dual_Dense_onlycil.py:592-628): 320x384 grid whose obstacle crosses
    the hard-wired row 200; (dp/dx, dp/dy) = gradient of an analytic p + noise; sdfunct in metres (< 1)."""
    Ny, Nx, delta = 320, 384, 5e-3
    g = synthetic.channel_grid(Ny, Nx, seed=41, cx=0.35, cy=200.5 / Ny, r=0.1)
    sdfunct = g[..., 2] * 0.3
    x_min, y_min = -0.5, -0.8
    x_max, y_max = x_min + Nx * delta, y_min + Ny * delta
    X0 = np.linspace(x_min + delta / 2, x_max - delta / 2, Nx)
    yy, xx = np.meshgrid(np.linspace(y_min, y_max, Ny), np.linspace(x_min, x_max, Nx), indexing="ij")
    rng = np.random.default_rng(5)
    gx = 3 * np.cos(3 * xx) * np.cos(2 * yy) + 0.05 * rng.standard_normal((Ny, Nx))
    gy = -2 * np.sin(3 * xx) * np.sin(2 * yy) + 0.05 * rng.standard_normal((Ny, Nx))
    gradP = np.stack([gx, gy], -1)
    gradP[sdfunct == 0] = 0.0
    return dict(gradP=gradP, sdfunct=sdfunct, X0=X0, delta=delta, min_x=x_min, max_x=x_max, min_y=y_min, max_y=y_max)


def build_poisson_case():
    """pressureSM_Poisson feature builder (SM_call.py:588-711): dimensional velocity / delta-velocity grids
    (zero outside the flow), raw SDF image, scales."""
    Ny, Nx = 160, 200
    g = synthetic.channel_grid(Ny, Nx, seed=51, obstacle="circle")
    d = synthetic.delta_grid(Ny, Nx, seed=52, step=3)
    sdf = g[..., 2] * 0.3
    ux, uy = 1.3 * g[..., 0], 1.3 * g[..., 1]
    dux, duy = 0.05 * d[..., 0], 0.05 * d[..., 1]
    for a in (ux, uy, dux, duy):
        a[sdf == 0] = 0.0
    U = float(np.sqrt(ux ** 2 + uy ** 2).max())
    return dict(ux=ux, uy=uy, dux=dux, duy=duy, sdfunct=sdf, L=0.25, U=U, k=0.5,
                max_abs=(2.7, 0.031, 0.027, 0.29))


def build_domain_case(shape: str = "circle"):
    """Inputs of `domain_dist` (python_module.py:72-99): `top` patch points (both channel walls), obstacle boundary
    points, and the target points = a uniform 5e-3 grid plus the hard cases of a point-in-polygon test -- hull vertices,
    edge mid-points, points level with a vertex, points on the bounding box.  'rectangle': obstacle edges that run
    exactly through grid points."""
    from psm_amd import geometry
    delta = 5e-3
    X0, Y0 = geometry.create_uniform_grid(0.0, 1.5, -0.35, 0.35, delta)
    xs, ys = np.unique(X0), np.unique(Y0)
    wall = np.linspace(0.0, 1.5, 301)
    top = np.concatenate([np.c_[wall, np.full_like(wall, 0.35)], np.c_[wall, np.full_like(wall, -0.35)]])
    if shape == "circle":
        th = np.linspace(0.0, 2 * np.pi, 90, endpoint=False)
        obst = np.c_[0.4 + 0.1 * np.cos(th), 0.013 + 0.1 * np.sin(th)]
    else:                                                     # corners ON grid points
        x0, x1, y0, y1 = xs[60], xs[80], ys[59], ys[79]
        ex, ey = np.linspace(x0, x1, 21), np.linspace(y0, y1, 21)
        obst = np.concatenate([np.c_[ex, np.full_like(ex, y0)], np.c_[ex, np.full_like(ex, y1)],
                               np.c_[np.full_like(ey, x0), ey], np.c_[np.full_like(ey, x1), ey]])
    ring = geometry.convex_hull_ring(obst)
    mids = 0.5 * (ring[:-1] + ring[1:])
    level = np.c_[np.repeat([0.05, 0.4, 0.9], len(ring) - 1), np.tile(ring[:-1, 1], 3)]       # rays through vertices
    boxpts = np.array([[0.0, 0.35], [1.5, -0.35], [0.75, 0.35], [0.0, 0.0], [1.5, 0.1], [-0.001, 0.0], [0.7, 0.3501]])
    xy0 = np.concatenate([np.c_[X0, Y0], ring[:-1], mids, level, boxpts])
    return top, obst, xy0


def build_idw_case():
    """Inputs of the Improved_SM `interp_weights` (pressureSM_deltas/utils.py:22-55): a scattered source cloud and a
    target lattice that reaches beyond its hull on every side (the IDW fallback branch)."""
    rng = np.random.default_rng(909)
    xyz = np.c_[rng.random(1500) * 1.0, rng.random(1500) * 0.6 - 0.3]
    gx, gy = np.meshgrid(np.linspace(-0.05, 1.05, 89), np.linspace(-0.34, 0.34, 55))
    return xyz, np.c_[gx.ravel(), gy.ravel()]


DATASET_MAXS = (0.062, 0.055, 0.31, 0.047)


POISSON_MAXS = (1.7, 0.062, 0.055, 0.31, 0.047)


def build_dataset_case(directory: str, poisson: bool = False):
    """Everything `Evaluation(delta, shape, overlap, var_p, var_in, dataset_path, model_path, ...)` reads from
    disk, written into `directory` in the reference's formats (tests/h5write.py for HDF5): the padded
    dataset (1 sim x 3 frames, 11 columns, pad -100), `maxs`, pickled scikit-learn PCA objects, `mean_std.npz`
    and a Keras-style Dense `.h5`.  Returns the in-memory truth for the oracle."""
    import pickle
    from sklearn.decomposition import PCA
    import h5write
    frames = [synthetic.channel_mesh(step=s) for s in range(4)]
    top, obst = frames[0][1], frames[0][2]
    N = frames[0][0].shape[0]
    T, max_cells, max_pts = 3, N + 41, max(len(top), len(obst)) + 13
    sim = np.full((1, T, max_cells, 11), -100.0, np.float32)
    for t in range(T):
        cur, prev = frames[t + 1][0], frames[t][0]
        pp = frames[t - 1][0] if t > 0 else frames[0][0] * 0.999
        sim[0, t, :N, 0:2] = cur[:, 0:2]; sim[0, t, :N, 2] = cur[:, 4]; sim[0, t, :N, 3:5] = cur[:, 2:4]
        sim[0, t, :N, 5:7] = cur[:, 0:2] - prev[:, 0:2]; sim[0, t, :N, 7] = cur[:, 4] - prev[:, 4]
        sim[0, t, :N, 8:10] = prev[:, 0:2] - pp[:, 0:2]; sim[0, t, :N, 10] = prev[:, 4] - pp[:, 4]
    tb = np.full((1, T, max_pts, 2), -100.0, np.float32); ob = tb.copy()
    tb[0, :, :len(top)] = top; ob[0, :, :len(obst)] = obst
    h5write.write_h5(os.path.join(directory, "dataset.hdf5"), {"sim_data": sim, "top_bound": tb, "obst_bound": ob})
    np.savetxt(os.path.join(directory, "maxs"), np.array(POISSON_MAXS if poisson else DATASET_MAXS))
    P, PC = 32, 24                                               # stored components, components the rule keeps
    cin = 4 if poisson else 3
    full = synthetic.make_model("deltas", p_in=P, p_out=P, seed_pca=2024, seed_w=3, scaler_kind="std", c_in=cin)
    model = synthetic.make_model("deltas", p_in=PC, p_out=PC, seed_pca=2024, seed_w=3, scaler_kind="std", c_in=cin)
    model.sdf_ch = 3 if poisson else 2
    model.comp_in, model.comp_out = full.comp_in[:PC], full.comp_out[:PC]
    model.mean_in, model.mean_out = full.mean_in, full.mean_out
    evr = np.array([0.0395] * PC + [0.004] * (P - PC))            # cumulative ratio first exceeds 0.95 at index 24
    for stem, comp, mean in (("ipca_input", full.comp_in, full.mean_in), ("ipca_p", full.comp_out, full.mean_out)):
        o = PCA(n_components=P)
        o.components_, o.mean_, o.explained_variance_ratio_ = comp, mean, evr
        o.explained_variance_, o.n_components_, o.n_features_in_ = evr.copy(), P, comp.shape[1]
        with open(os.path.join(directory, stem + ".pkl"), "wb") as f:
            pickle.dump(o, f)
    np.savez(os.path.join(directory, "mean_std.npz"), mean_in=model.in_a, std_in=model.in_b, mean_out=model.out_a, std_out=model.out_b)
    h5write.write_keras_dense(os.path.join(directory, "model.h5"), model.weights)
    return dict(sim=sim, top=top, obst=obst, N=N, model=model, dataset_path=os.path.join(directory, "dataset.hdf5"),
                model_path=os.path.join(directory, "model.h5"))


GRADP_MAXS = (1.0, 0.41, 0.33, 2.9, 2.4)


def build_gradp_dataset_case(directory: str):
    """Artefact directory of the U_to_gradP evaluator (Eval_dual_Dense_onlycil.py:30-66, 160-253, 418-640): a 320x300
    grid whose obstacle crosses the hard-wired row 200; dataset columns 0 Ux, 1 Uy, 2 p, 3 Cx, 4 Cy, 5 (test column),
    6 dP/dx, 7 dP/dy; `maxs` (5 values), `maxs_PCA`, pickled PCA objects, Keras-style `.h5`."""
    import pickle
    from sklearn.decomposition import PCA
    import h5write
    T = 2
    frames = [synthetic.channel_mesh(Lx=1.5, Ly=1.6, h=0.008, cy=0.2, R=0.1, step=s) for s in range(T)]
    top, obst = frames[0][1], frames[0][2]
    N = frames[0][0].shape[0]
    max_cells, max_pts = N + 29, max(len(top), len(obst)) + 7
    sim = np.full((1, T, max_cells, 8), -100.0, np.float32)
    for t in range(T):
        a = frames[t][0]
        X, Y = a[:, 2], a[:, 3]
        sim[0, t, :N, 0:2] = a[:, 0:2]; sim[0, t, :N, 2] = a[:, 4]; sim[0, t, :N, 3:5] = a[:, 2:4]
        sim[0, t, :N, 5] = np.cos(3 * X) * np.sin(2 * Y)
        sim[0, t, :N, 6] = -0.4 / 1.5 - 0.5 * np.sin(5 * X + 0.15 * t) * (2 * Y / 1.6)      # d/dx of channel_mesh's p
        sim[0, t, :N, 7] = 0.1 * np.cos(5 * X + 0.15 * t) * (2 / 1.6)                       # d/dy
    tb = np.full((1, T, max_pts, 2), -100.0, np.float32); ob = tb.copy()
    tb[0, :, :len(top)] = top; ob[0, :, :len(obst)] = obst
    h5write.write_h5(os.path.join(directory, "dataset.hdf5"), {"sim_data": sim, "top_bound": tb, "obst_bound": ob})
    np.savetxt(os.path.join(directory, "maxs"), np.array(GRADP_MAXS))
    P, PC = 32, 24
    full = synthetic.make_model("gradp", p_in=P, p_out=P, seed_pca=3030, seed_w=9)
    model = synthetic.make_model("gradp", p_in=PC, p_out=PC, seed_pca=3030, seed_w=9)
    model.comp_in, model.comp_out = full.comp_in[:PC], full.comp_out[:PC]
    model.mean_in, model.mean_out = full.mean_in, full.mean_out
    np.savetxt(os.path.join(directory, "maxs_PCA"), np.array([model.in_a, model.out_a]))
    evr = np.array([0.0395] * PC + [0.004] * (P - PC))
    for stem, comp, mean in (("ipca_input", full.comp_in, full.mean_in), ("ipca_p", full.comp_out, full.mean_out)):
        o = PCA(n_components=P)
        o.components_, o.mean_, o.explained_variance_ratio_ = comp, mean, evr
        with open(os.path.join(directory, stem + ".pkl"), "wb") as f:
            pickle.dump(o, f)
    h5write.write_keras_dense(os.path.join(directory, "model.h5"), model.weights)
    return dict(sim=sim, top=top, obst=obst, N=N, model=model, dataset_path=os.path.join(directory, "dataset.hdf5"),
                model_path=os.path.join(directory, "model.h5"))
