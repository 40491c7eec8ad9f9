"""Seeded synthetic inputs and models for the BASELINE.json configs (SURVEY.md §8 d).

There is no network for datasets or checkpoints and the reference ships no PCA
pickles (.MISSING_LARGE_BLOBS:27-31), so every benchmark / parity input is
generated here from fixed seeds: normalised ``grid[Ny,Nx,C]`` images of the
shape the reference builds at python_module.py:288-297 / SM_call.py:430-444 /
Eval_dual_Dense_onlycil.py:453-468, orthonormal PCA bases standing in for the
sklearn ``components_`` / ``mean_`` and He-initialised dense stacks with the
reference architectures (utils.py:435-461 ``define_model_arch``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, List, Sequence, Tuple

import numpy as np

ARCHS = {  # utils.py:435-461
    "MLP_small": [512] * 3,
    "MLP_big": [256] + [512] * 5 + [256],
    "MLP_huge": [256] + [512] * 10 + [256],
    "MLP_huger": [256] + [512] * 18 + [256],
    "MLP_small_unet": [512, 256, 128, 64, 32, 64, 128, 256, 512],
    "MLP_attention": [512] * 3,       # densePCA_attention: the same three Dense layers + attention / LayerNormalization
}


@dataclass
class SurrogateModel:
    """Host-side description of one trained surrogate (the artefacts the
    reference loads at python_module.py:103-118,168-170 / SM_call.py:70-87)."""
    variant: str                  # 'chapter5' | 'deltas' | 'gradp'
    c_in: int
    c_out: int
    comp_in: np.ndarray           # [P_i, S*S*c_in]  f64
    mean_in: np.ndarray           # [S*S*c_in]
    comp_out: np.ndarray          # [P_o, S*S*c_out]
    mean_out: np.ndarray          # [S*S*c_out]
    weights: List[Tuple[np.ndarray, np.ndarray]]   # Keras layout W[in,out], b[out], f32
    scaler_kind: str = "max_abs"  # 'max_abs' | 'std' | 'min_max'
    in_a: np.ndarray | float = 1.0
    in_b: np.ndarray | float = 1.0
    out_a: np.ndarray | float = 1.0
    out_b: np.ndarray | float = 1.0
    out_scale: float = 1.0
    S: int = 128
    ov: int | None = None
    sdf_ch: int = 2
    # conv1D_PCA head (NNs.py:75-124): Conv1D layers [(kernel[k, c_in, c_out] f32, bias[c_out] f32)] in front of `weights`
    conv1d: List[Tuple[np.ndarray, np.ndarray]] = field(default_factory=list)
    # densePCA_attention (NNs.py:40-72, 'MLP_attention'): `weights` = the n_layers Dense layers + the head; this dict holds the
    # MultiHeadAttention block behind the first of them -- Wq, bq, Wk, bk, Wv, bv [d, heads, dim] / [heads, dim], Wo [heads, dim, d],
    # bo [d] (Keras EinsumDense layouts) -- and "ln": the n_layers LayerNormalizations [(gamma[d], beta[d])], "eps"
    attention: Optional[dict] = None

    @property
    def p_in(self):
        return self.comp_in.shape[0]

    @property
    def p_out(self):
        return self.comp_out.shape[0]


def orthonormal_basis(P: int, K: int, seed: int) -> np.ndarray:
    """[P, K] with orthonormal rows (QR of a seeded Gaussian), like sklearn's
    ``components_``."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((K, P))
    q, _ = np.linalg.qr(g)
    return np.ascontiguousarray(q.T)


def he_dense_stack(p_in: int, widths: Sequence[int], p_out: int, seed: int):
    rng = np.random.default_rng(seed)
    dims = [p_in] + list(widths) + [p_out]
    out = []
    for a, b in zip(dims[:-1], dims[1:]):
        W = (rng.standard_normal((a, b)) * np.sqrt(2.0 / a)).astype(np.float32)
        bias = (rng.standard_normal(b) * 0.01).astype(np.float32)
        out.append((W, bias))
    return out


CONV1D_WIDTHS = [128, 64, 32, 16, 32, 64, 128]        # utils.define_model_arch('conv1D'), utils.py:452-454


def he_conv1d_head(p_in: int, filters: Sequence[int], p_out: int, seed: int, kernel_size: int = 3):
    """Seeded random-init weights of the reference's conv1D_PCA network -> (conv1d layers, dense layers)."""
    rng = np.random.default_rng(seed)
    convs, cin = [], 1
    for f in filters:
        K = (rng.standard_normal((kernel_size, cin, f)) * np.sqrt(2.0 / (kernel_size * cin))).astype(np.float32)
        convs.append((K, (rng.standard_normal(f) * 0.01).astype(np.float32)))
        cin = f
    n = p_in * cin
    W = (rng.standard_normal((n, p_out)) * np.sqrt(1.0 / n)).astype(np.float32)
    return convs, [(W, (rng.standard_normal(p_out) * 0.01).astype(np.float32))]


def he_attention_block(widths: Sequence[int], seed: int, n_heads: int = 8, key_dim: int = 64, eps: float = 1e-3) -> dict:
    """Seeded random-init parameters of the attention part of the reference's densePCA_attention (NNs.py:53-64):
    MultiHeadAttention(num_heads=8, key_dim=64) behind the first Dense layer and one LayerNormalization per layer.  gamma /
    beta are drawn away from their (1, 0) initial values, as trained ones would be."""
    rng = np.random.default_rng(seed)
    d = int(widths[0])
    if any(int(w) != d for w in widths):
        raise ValueError("densePCA_attention adds the attention output to every further layer: equal widths only")
    g = lambda *sh: (rng.standard_normal(sh) * np.sqrt(1.0 / sh[0])).astype(np.float32)
    att = {"Wq": g(d, n_heads, key_dim), "bq": (rng.standard_normal((n_heads, key_dim)) * 0.05).astype(np.float32),
           "Wk": g(d, n_heads, key_dim), "bk": (rng.standard_normal((n_heads, key_dim)) * 0.05).astype(np.float32),
           "Wv": g(d, n_heads, key_dim), "bv": (rng.standard_normal((n_heads, key_dim)) * 0.05).astype(np.float32),
           "Wo": (rng.standard_normal((n_heads, key_dim, d)) * np.sqrt(1.0 / (n_heads * key_dim))).astype(np.float32),
           "bo": (rng.standard_normal(d) * 0.05).astype(np.float32), "eps": float(eps),
           "ln": [((1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32), (0.05 * rng.standard_normal(d)).astype(np.float32))
                  for _ in widths]}
    return att


def make_model(variant: str, p_in: int = 128, p_out: int = 128, arch: str = "MLP_small",
               scaler_kind: str | None = None, weights=None, seed_pca: int = 1234,
               seed_w: int = 7, S: int = 128, c_in: int = 3, c_out: int | None = None,
               out_scale: float = 1.0) -> SurrogateModel:
    if c_out is None:
        c_out = 2 if variant == "gradp" else 1
    if scaler_kind is None:
        scaler_kind = "std" if variant == "deltas" else "max_abs"
    K_in, K_out = S * S * c_in, S * S * c_out
    rng = np.random.default_rng(seed_pca + 1)
    comp_in = orthonormal_basis(p_in, K_in, seed_pca)
    comp_out = orthonormal_basis(p_out, K_out, seed_pca + 7)
    mean_in = rng.standard_normal(K_in) * 0.05
    mean_out = rng.standard_normal(K_out) * 0.05
    if weights is None:
        weights = he_dense_stack(p_in, ARCHS[arch], p_out, seed_w)
    m = SurrogateModel(variant, c_in, c_out, comp_in, mean_in, comp_out, mean_out, list(weights),
                       scaler_kind=scaler_kind, out_scale=out_scale, S=S)
    if arch == "MLP_attention":
        m.attention = he_attention_block(ARCHS[arch], seed_w + 100)
    if scaler_kind == "max_abs":
        # like the reference's maxs_PCA file (147.2, 26.7) but matched to the magnitude of the
        # synthetic PCA coefficients so that the network input is O(1)
        m.in_a, m.out_a = 0.6, 12.0
    elif scaler_kind == "std":
        m.in_a = rng.standard_normal(p_in) * 0.1
        m.in_b = 0.3 + rng.random(p_in) * 0.6
        m.out_a = rng.standard_normal(p_out) * 0.3
        m.out_b = 2.0 + rng.random(p_out) * 4.0
    elif scaler_kind == "min_max":
        m.in_a = -1.5 - rng.random(p_in) * 0.5
        m.in_b = 1.5 + rng.random(p_in) * 0.5
        m.out_a = -6.0 - rng.random(p_out)
        m.out_b = 6.0 + rng.random(p_out)
    else:
        raise ValueError("Standardization method not valid")
    return m


# --------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------
def _wall_sdf(Ny, Nx, yy, xx, obst_mask, obst_dist, walls="channel"):
    """Distance to the nearest wall / obstacle, 0 inside solids, max-normalised."""
    if walls == "channel":
        d = np.minimum(yy + 0.5, Ny - 0.5 - yy)
    else:   # closed box
        d = np.minimum(np.minimum(yy + 0.5, Ny - 0.5 - yy), np.minimum(xx + 0.5, Nx - 0.5 - xx))
    if obst_dist is not None:
        d = np.minimum(d, obst_dist)
    d = np.where(obst_mask, 0.0, d)
    return d / d.max()


def cavity_grid(N: int = 128) -> np.ndarray:
    """BASELINE config 0: analytic cavity-like vortex, psi = sin^2(pi x) sin^2(pi y)."""
    yy, xx = np.meshgrid(np.arange(N, dtype=np.float64), np.arange(N, dtype=np.float64), indexing="ij")
    x, y = (xx + 0.5) / N, (yy + 0.5) / N
    u = np.sin(np.pi * x) ** 2 * 2 * np.pi * np.sin(np.pi * y) * np.cos(np.pi * y)
    v = -2 * np.pi * np.sin(np.pi * x) * np.cos(np.pi * x) * np.sin(np.pi * y) ** 2
    s = max(np.abs(u).max(), np.abs(v).max())
    sdf = _wall_sdf(N, N, yy, xx, np.zeros((N, N), bool), None, walls="box")
    return np.stack([u / s, v / s, sdf], axis=-1)


def channel_grid(Ny: int = 256, Nx: int = 256, seed: int = 1, noise: float = 0.02,
                 obstacle: str = "circle", cx: float = 0.3, cy: float = 0.5, r: float = 0.125,
                 extra_channels: int = 0) -> np.ndarray:
    """BASELINE configs 1/3/4: parabolic channel inflow with a potential-flow
    perturbation around an obstacle plus seeded noise; channels (Ux, Uy, SDF)
    scaled to |.|<=1, zero inside the obstacle.  ``extra_channels`` appends
    smooth label-like channels (used only by reassembly identity tests)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(Ny, dtype=np.float64), np.arange(Nx, dtype=np.float64), indexing="ij")
    X0, Y0, R = cx * Nx, cy * Ny, r * Ny
    dx, dy = xx + 0.5 - X0, yy + 0.5 - Y0
    if obstacle == "circle":
        rr = np.sqrt(dx * dx + dy * dy)
        inside = rr < R
        dist = np.maximum(rr - R, 0.0)
    elif obstacle == "rectangle":
        qx, qy = np.abs(dx) - R, np.abs(dy) - 0.6 * R
        inside = (qx < 0) & (qy < 0)
        dist = np.sqrt(np.maximum(qx, 0) ** 2 + np.maximum(qy, 0) ** 2)
        rr = np.maximum(np.sqrt(dx * dx + dy * dy), 1e-9)
    elif obstacle == "plate":
        qx, qy = np.abs(dx) - 0.15 * R, np.abs(dy) - R
        inside = (qx < 0) & (qy < 0)
        dist = np.sqrt(np.maximum(qx, 0) ** 2 + np.maximum(qy, 0) ** 2)
        rr = np.maximum(np.sqrt(dx * dx + dy * dy), 1e-9)
    elif obstacle == "none":
        inside = np.zeros((Ny, Nx), bool)
        dist = None
        rr = np.maximum(np.sqrt(dx * dx + dy * dy), 1e-9)
    else:
        raise ValueError(obstacle)
    yn = (yy + 0.5) / Ny * 2 - 1
    base = 1.5 * (1 - yn * yn)
    rr = np.maximum(rr, 1e-9)
    k = (R / rr) ** 2
    cos2, sin2 = (dx * dx - dy * dy) / (rr * rr), 2 * dx * dy / (rr * rr)
    u = base * (1 - k * cos2) + noise * rng.standard_normal((Ny, Nx))
    v = base * (-k * sin2) + noise * rng.standard_normal((Ny, Nx))
    u[inside] = 0.0
    v[inside] = 0.0
    u /= np.abs(u).max()
    v /= max(np.abs(v).max(), 1e-12)
    sdf = _wall_sdf(Ny, Nx, yy, xx, inside, dist)
    chans = [u, v, sdf]
    for e in range(extra_channels):
        ph = 0.7 * (e + 1)
        lab = np.sin(2 * np.pi * xx / Nx * (1 + e) + ph) * np.cos(np.pi * yy / Ny * (2 + e)) + 0.3 * (xx / Nx)
        lab[inside] = 0.0
        chans.append(lab)
    return np.stack(chans, axis=-1)


def delta_grid(Ny: int = 256, Nx: int = 256, seed: int = 2, step: int = 0, **kw) -> np.ndarray:
    """BASELINE config 2: dU = field(t) - field(t-1) of the channel field advected
    by a seeded phase shift per step; (dUx, dUy, SDF)."""
    rng = np.random.default_rng(seed + 7919 * step)
    g0 = channel_grid(Ny, Nx, seed=seed, **kw)
    sh = int(rng.integers(1, 6))
    g1 = g0.copy()
    g1[..., :2] = np.roll(g0[..., :2], sh, axis=1) * (1.0 + 0.05 * rng.standard_normal())
    d = g1[..., :2] - g0[..., :2]
    solid = g0[..., 2] == 0
    d[solid] = 0.0
    d /= max(np.abs(d).max(), 1e-12)
    return np.concatenate([d, g0[..., 2:3]], axis=-1)


def random_obstacle_cases(n: int, Ny: int = 256, Nx: int = 256, seed: int = 3) -> np.ndarray:
    """BASELINE config 3: ``n`` independent random-obstacle cases [n,Ny,Nx,3]."""
    rng = np.random.default_rng(seed)
    shapes = ("circle", "rectangle", "plate")
    out = np.empty((n, Ny, Nx, 3))
    for i in range(n):
        out[i] = channel_grid(Ny, Nx, seed=int(rng.integers(1 << 30)), obstacle=shapes[i % 3],
                              cx=float(rng.uniform(0.2, 0.6)), cy=float(rng.uniform(0.35, 0.65)),
                              r=float(rng.uniform(0.06, 0.14)))
    return out


# --------------------------------------------------------------------------
# solver-side input: unstructured cell centres + boundary patches (PythonComm_init.H:58-77)
# --------------------------------------------------------------------------
def channel_mesh(Lx: float = 1.5, Ly: float = 0.7, h: float = 0.008, seed: int = 5, cx: float = 0.4, cy: float = 0.0,
                 R: float = 0.08, step: int = 0):
    """Jittered cell-centre cloud of a channel with a circular obstacle.
    Returns (array[N,5] = Ux, Uy, Cx, Cy, p; top[Nt,2] (both walls); obst[No,2])."""
    rng = np.random.default_rng(seed)
    nx, ny = int(round(Lx / h)), int(round(Ly / h))
    X, Y = np.meshgrid((np.arange(nx) + 0.5) * h, (np.arange(ny) + 0.5) * h - Ly / 2)
    X = X + rng.uniform(-0.25, 0.25, X.shape) * h
    Y = Y + rng.uniform(-0.25, 0.25, Y.shape) * h
    X, Y = X.ravel(), Y.ravel()
    keep = (X - cx) ** 2 + (Y - cy) ** 2 > (R + 0.3 * h) ** 2
    X, Y = X[keep], Y[keep]
    yn = 2 * Y / Ly
    rr2 = np.maximum((X - cx) ** 2 + (Y - cy) ** 2, 1e-12)
    k = R * R / rr2
    dx, dy = X - cx, Y - cy
    ph = 0.15 * step
    Ux = 1.2 * (1 - yn * yn) * (1 - k * (dx * dx - dy * dy) / rr2) + 0.03 * np.sin(9 * X + ph) * np.cos(7 * Y)
    Uy = 1.2 * (1 - yn * yn) * (-k * 2 * dx * dy / rr2) + 0.03 * np.cos(8 * X - ph) * np.sin(6 * Y)
    p = 0.4 * (Lx - X) / Lx + 0.1 * np.cos(5 * X + ph) * yn
    array = np.stack([Ux, Uy, X, Y, p], axis=1)
    xs = np.arange(0.0, Lx + 1e-9, h / 2)
    top = np.concatenate([np.stack([xs, np.full_like(xs, Ly / 2)], 1), np.stack([xs, np.full_like(xs, -Ly / 2)], 1)])
    th = np.linspace(0, 2 * np.pi, 240, endpoint=False)
    obst = np.stack([cx + R * np.cos(th), cy + R * np.sin(th)], 1)
    return array, top, obst


def shipped_case_mesh(delta: float = 0.005, ny: int = 400, nx: int = 3000, h: float = 0.031):
    """A channel mesh whose cell-centre extents are exactly ny * delta by nx * delta, so that init_func's grid is the reference's shipped
    case shape (400 x 3000 at delta 0.005: 104 blocks of the Chapter-5 layout; python_module.py:190-217): channel_mesh stretched by a
    fraction of a percent in x and y (its jitter leaves the extents a few cells short: 400 x 2998, an unaligned row pitch)."""
    array, top, obst = channel_mesh(Lx=nx * delta, Ly=ny * delta, h=h, cx=3.0, R=0.25)
    x0, x1, y0, y1 = array[:, 2].min(), array[:, 2].max(), array[:, 3].min(), array[:, 3].max()
    sx, sy = nx * delta / (x1 - x0), ny * delta / (y1 - y0)
    ym = 0.5 * (y0 + y1)
    def tf(pts, cx, cy):
        pts = pts.copy()
        pts[:, cx] = (pts[:, cx] - x0) * sx + x0
        pts[:, cy] = (pts[:, cy] - ym) * sy + ym
        return pts
    return tf(array, 2, 3), tf(top, 0, 1), tf(obst, 0, 1)


# --------------------------------------------------------------------------
# convolutional path: layer shapes and seeded weights of the build-defined UNet-S (include/psm_unet.h)
# --------------------------------------------------------------------------
UNET_WIDTHS_S = (16, 32, 64, 128, 256)


def unet_conv_shapes(c_in: int = 3, widths=UNET_WIDTHS_S, c_out: int = 1):
    """[(k, c_in, c_out)] in the order psm_unet_set_conv expects: enc0a, enc0b, ..., dec0a, dec0b, head."""
    L = len(widths)
    out = []
    for l in range(L):
        out += [(3, c_in if l == 0 else widths[l - 1], widths[l]), (3, widths[l], widths[l])]
    for l in range(L - 2, -1, -1):
        out += [(3, widths[l + 1] + widths[l], widths[l]), (3, widths[l], widths[l])]
    out.append((1, widths[0], c_out))
    return out


def unet_he_weights(c_in: int = 3, widths=UNET_WIDTHS_S, c_out: int = 1, seed: int = 7):
    """Seeded He-normal Conv2D kernels [k,k,c_in,c_out] and small biases, float32 (random-init weights of the
    architecture: there is no trained U-Net anywhere in the reference)."""
    rng = np.random.default_rng(seed)
    out = []
    for k, ci, co in unet_conv_shapes(c_in, widths, c_out):
        W = (rng.standard_normal((k, k, ci, co)) * np.sqrt(2.0 / (k * k * ci))).astype(np.float32)
        b = (rng.standard_normal(co) * 0.05).astype(np.float32)
        out.append((W, b))
    return out
