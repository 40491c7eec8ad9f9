"""Randomised soak of the convolutional path against the NumPy U-Net oracle (run by hand on a GPU box; tests/test_unet.py
holds the fixed cases).  Every trial draws a depth, level widths, channel counts, an image shape (a multiple of 2^(levels-1),
often not a multiple of the 16-pixel / 30 x 14 tiles), a case count, the precision and whether the plan is autotuned, and
compares the field with the oracle's forward pass under the same rounding points.

    python tests/measure/soak_unet.py [trials] [seed]
"""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import psm_amd
from psm_amd import UNetSurrogate
from oracle import unet_oracle as uo

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rng = np.random.default_rng(seed)
worst = {"f32": 0.0, "bf16": 0.0}
n_pairs = n_split = n_x6 = 0
t0 = time.time()
for trial in range(trials):
    levels = int(rng.integers(2, 6))
    w0 = int(rng.choice([16, 16, 32, 48, 64]))
    widths = [w0]
    for _ in range(levels - 1):
        widths.append(int(min(320, 16 * max(1, round(widths[-1] * float(rng.choice([1.0, 1.5, 2.0])) / 16)))))
    widths = tuple(widths)
    c_in, c_out = int(rng.integers(1, 8)), int(rng.integers(1, 5))
    m = 1 << (levels - 1)
    ny, nx = m * int(rng.integers(1, max(2, 400 // m))), m * int(rng.integers(1, max(2, 400 // m)))
    if rng.random() < 0.2:
        ny, nx = 256, 256
    n = int(rng.integers(1, 5)) if ny * nx <= 160 * 160 else int(rng.integers(1, 3))
    prec = "bf16" if rng.random() < 0.5 else "f32"
    tune = bool(rng.random() < 0.25)
    specs = uo.unet_specs(c_in, widths, c_out)
    W = uo.he_weights(specs, seed=int(rng.integers(1 << 20)))
    grids = rng.standard_normal((n, ny, nx, c_in)).astype(np.float32)
    info = dict(trial=trial, widths=widths, c_in=c_in, c_out=c_out, ny=ny, nx=nx, n=n, precision=prec, autotune=tune)
    with UNetSurrogate(W, ny, nx, c_in=c_in, c_out=c_out, widths=widths, max_cases=n, precision=prec, autotune=tune) as net:
        out = net.forward(grids)
        plan = [net.plan_info(i) for i in range(len(specs))]
    n_pairs += any(p[3] & 3 for p in plan)
    n_x6 += any(p[3] & 4 for p in plan)
    n_split += any(p[2] > 1 for p in plan)
    for k in range(n):
        ref = uo.unet_forward(grids[k], W, widths, precision=prec) if prec == "bf16" else uo.unet_forward(grids[k], W, widths)
        den = max(float(np.abs(ref).max()), 1e-6)
        if prec == "bf16":
            err = float(np.linalg.norm(out[k].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-12))
            tol = 2e-2
        else:
            err = float(np.abs(out[k] - ref).max() / den)
            tol = 1e-4
        worst[prec] = max(worst[prec], err)
        if not np.isfinite(err) or err > tol:
            raise SystemExit(f"mismatch {err:.2e} > {tol}: {info} case {k} plan {plan}")
    if trial % 10 == 9:
        print(f"trial {trial + 1}/{trials}: worst f32 max-abs/max {worst['f32']:.2e}, worst bf16 rel-L2 {worst['bf16']:.2e}, "
              f"{n_pairs} with fused pairs, {n_split} with split-K, {n_x6} with x6 layers, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK OK: {trials} networks, worst f32 {worst['f32']:.2e} (tol 1e-4), worst bf16 rel-L2 {worst['bf16']:.2e} (tol 2e-2), "
      f"{n_pairs} with fused pairs, {n_split} with split-K, {n_x6} with x6 layers, seed {seed}")
