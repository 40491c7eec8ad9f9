"""Randomised soak of the rows SURVEY.md section 8 marks "next" against their CPU restatements (run by hand on a GPU box;
tests/ holds the fixed cases): integration of (dp/dx, dp/dy) into p with the four-quadrant stitching
(Eval_dual_Dense_onlycil.py:371-416, 592-628), scipy.ndimage.gaussian_filter of the post-steps (SM_call.py:353-363) and the
pressureSM_Poisson feature image (SM_call.py:588-711).  Every trial draws a grid shape, an obstacle (or none) plus solid
bands, the cut of the quadrants anywhere in the grid, cell sizes, filter widths and feature scales.

    python tests/measure/soak_next_rows.py [trials] [seed]
"""
import os, sys, time
import numpy as np
import scipy.ndimage as ndi
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psm_amd import GridSurrogate, synthetic, _lib
from oracle import psm_oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 77
rng = np.random.default_rng(seed)
worst = {"integrate": 0.0, "gauss": 0.0, "features": 0.0}
TOL = {"integrate": 2e-4, "gauss": 2e-5, "features": 2e-4}
t0 = time.time()
n_nan = 0
n_undef = 0


def check(tag, got, want, info, scale=None):
    global n_nan
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        raise SystemExit(f"NaN pattern differs: {tag} {info}: got {int(np.isnan(got).sum())}, oracle {int(np.isnan(want).sum())}")
    ok = ~np.isnan(want)
    n_nan += int((~ok).any())
    if ok.any():
        s = scale if scale is not None else max(1e-6, float(np.abs(want[ok]).max()))
        err = float(np.abs(got[ok] - want[ok]).max() / s)
        worst[tag] = max(worst[tag], err)
        if err > TOL[tag]:
            raise SystemExit(f"mismatch {err:.2e} > {TOL[tag]}: {tag} {info}")


model = synthetic.make_model("gradp", p_in=4, p_out=4)
m4 = synthetic.make_model("deltas", p_in=4, p_out=4, c_in=4); m4.sdf_ch = 3
for trial in range(trials):
    ny, nx = int(rng.integers(130, 420)), int(rng.integers(130, 640))
    obstacle = ("circle", "rectangle", "plate", "none")[int(rng.integers(4))]
    g = synthetic.channel_grid(ny, nx, seed=int(rng.integers(1 << 30)), obstacle=obstacle, cx=float(rng.uniform(0.15, 0.85)),
                               cy=float(rng.uniform(0.15, 0.85)), r=float(rng.uniform(0.04, 0.2)))
    for _ in range(int(rng.integers(0, 3))):                           # solid bands: empty columns / rows at the cut, resets
        h, w = int(rng.integers(2, 50)), int(rng.integers(2, 120))
        y0, x0 = int(rng.integers(0, ny - h)), int(rng.integers(0, nx - w))
        g[y0:y0 + h, x0:x0 + w, :] = 0.0
    sd = g[..., 2] * float(rng.uniform(0.05, 1.0))                     # metres (< 1) like the reference's, or up to exactly 1.0
    if rng.random() < 0.2:
        sd = sd / max(sd.max(), 1e-9)                                  # max == 1.0: int(sdfunct) == 1 somewhere (the index quirk)
    info = dict(trial=trial, ny=ny, nx=nx, obstacle=obstacle)
    # ---- integration
    cy, cx = int(rng.integers(1, ny - 1)), int(rng.integers(1, nx - 1))
    dx, dy = float(rng.uniform(1e-3, 2e-2)), float(rng.uniform(1e-3, 2e-2))
    gradp = rng.standard_normal((ny, nx, 2)) * 0.3 + np.stack([np.cos(np.arange(nx) * 0.02)[None, :] * np.ones((ny, 1)),
                                                                np.sin(np.arange(ny) * 0.03)[:, None] * np.ones((1, nx))], -1)
    gradp[sd == 0] = 0.0
    gradp = gradp.astype(np.float32)
    with GridSurrogate(model, ny, nx) as sur:
        try:
            sur.set_integration(sd, cy, cx, dx, dy)
            refused = False
        except _lib.PsmError as e:
            # geometries on which the reference itself raises are refused up front (psm_set_integration says which): the two
            # columns at the cut hold different numbers of flow cells -- the stitching subtracts two boolean-indexed columns of
            # different lengths (UGP:612, 624) --, or int(sdfunct) indexes outside a one-column quadrant (UGP:385-390)
            if "the reference raises" not in str(e):
                raise
            refused = True
        if refused:
            try:
                with np.errstate(all="ignore"):
                    orc.integrate_gradp(gradp.astype(np.float64), sd, dx, dy, cy, cx)
            except (ValueError, IndexError):
                n_undef += 1
            else:
                raise SystemExit(f"the library refused a cut the oracle integrates: {dict(info, cy=cy, cx=cx)}")
        else:
            got = sur.integrate_gradp(gradp)
            with np.errstate(all="ignore"):
                want = orc.integrate_gradp(gradp.astype(np.float64), sd, dx, dy, cy, cx)
            check("integrate", got, want, dict(info, cy=cy, cx=cx))
        # ---- gaussian filter (any 2-D field; sigma per axis)
        f = rng.standard_normal((ny, nx)).astype(np.float32)
        sig = (float(rng.uniform(0.5, 60.0)), float(rng.uniform(0.5, 60.0)))
        check("gauss", sur.gaussian_filter(f, sig), ndi.gaussian_filter(f.astype(np.float64), sigma=sig, order=0), dict(info, sigma=sig),
              scale=1.0)
    # ---- Poisson feature image
    d = synthetic.delta_grid(ny, nx, seed=int(rng.integers(1 << 30)), step=int(rng.integers(1, 6)))
    ux, uy = float(rng.uniform(0.3, 3.0)) * g[..., 0], float(rng.uniform(0.3, 3.0)) * g[..., 1]
    dux, duy = 0.05 * d[..., 0], 0.05 * d[..., 1]
    for a in (ux, uy, dux, duy):
        a[sd == 0] = 0.0
    U = float(np.sqrt(ux ** 2 + uy ** 2).max())
    if U == 0.0:
        continue
    args = (ux, uy, dux, duy, sd, float(rng.uniform(0.05, 1.0)), U, float(rng.uniform(0.2, 3.0)),
            tuple(float(v) for v in rng.uniform(0.02, 3.0, 4)))
    with GridSurrogate(m4, ny, nx) as sur:
        got = sur.poisson_features(*args)
        with np.errstate(all="ignore"):
            want = orc.poisson_features(*args)[0]
        check("features", got, want, info, scale=max(1.0, float(np.nanmax(np.abs(want)))))
    if trial % 10 == 9:
        print(f"trial {trial + 1}/{trials}: worst integrate {worst['integrate']:.2e}, gauss {worst['gauss']:.2e}, features {worst['features']:.2e}, "
              f"{n_nan} comparisons with NaN, {n_undef} cuts the reference cannot stitch, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK OK: {trials} draws ({n_undef} cuts refused by the reference and the library alike); worst rel err integrate {worst['integrate']:.2e} (tol {TOL['integrate']}), gaussian filter {worst['gauss']:.2e} "
      f"(tol {TOL['gauss']}), Poisson features {worst['features']:.2e} (tol {TOL['features']}), seed {seed}")
