"""CPU oracle for the pressure-surrogate hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement (written from the behaviour of the reference,
not copied from it) of the per-timestep surrogate pipeline

    grid -> overlapping 128x128 blocks -> PCA encode -> dense MLP -> PCA decode
         -> serial block-offset reassembly -> global reference shift

as performed by the three reference variants.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path (the HIP library behind ``include/psm.h``) never
does and fails loudly when the HIP extension is missing.

Reference files followed (paths relative to /root/reference):

  PM  = Thesis_Work/Chapter5/parallelized/test_case/python_module.py
  SMD = Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/SM_call.py
  UGP = Improved_SM/U_to_gradP/evaluation/Eval_dual_Dense_onlycil.py
  UTL = Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py
  NNS = Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/NNs.py

Parity status: PINNED for the block layout, PCA encode/decode and the
reassembly of all three variants by golden vectors produced by executing the
reference's own (pure-NumPy) statements in this container
(tests/golden/make_golden.py extracts them from the reference files at run
time; nothing of the reference is stored in this repository).  The dense
network itself is third-party arithmetic (Keras ``Dense`` == relu(x@W+b),
TensorFlow is not installable here) and is pinned only through the real
trained weights file and the published definition of the layer.

Precision mirrors the reference: float64 for PCA and reassembly, float32 for
the MLP (Keras floatx).

Build-defined behaviour (the reference is undefined there):
when the last block row coincides with the previous one (``p_i == 0``, e.g. a
256-row grid with the gradP stride of 32) the reference evaluates an empty
slice (NaN poisons the whole field, UGP:340/359) or raises a broadcast error
(SMD:335).  With ``degenerate='skip'`` (default) that duplicate row is left out
of the reassembly; ``degenerate='strict'`` reproduces the NumPy semantics.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

CHAPTER5 = "chapter5"
DELTAS = "deltas"
GRADP = "gradp"
VARIANTS = (CHAPTER5, DELTAS, GRADP)


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def mmean(values: np.ndarray, mask: np.ndarray) -> float:
    """Mean of ``values`` where ``mask`` is set; NaN for an empty selection
    (NumPy's behaviour for ``np.mean`` of an empty array, which every variant
    relies on and then probes with ``np.isnan``: SMD:252, UGP:315, PM:417)."""
    sel = values[mask]
    if sel.size == 0:
        return float("nan")
    return float(np.mean(sel))


def default_overlap(variant: str, S: int = 128) -> int:
    """PM:304 ``int(0.1*shape)``; SMD:788 / entry_point default 0.25;
    UGP:708 ``int(0.75*shape)`` (there called ``avance``)."""
    return {CHAPTER5: int(0.1 * S), DELTAS: int(0.25 * S), GRADP: int(0.75 * S)}[variant]


# --------------------------------------------------------------------------
# a7  block layout
# --------------------------------------------------------------------------
@dataclass
class Layout:
    variant: str
    Ny: int
    Nx: int
    S: int
    ov: int                      # width of the overlap strip ("avance" in PM/UGP)
    n_x: int
    n_y: int
    origins: List[Tuple[int, int]] = field(default_factory=list)   # (y0, x0)
    tags: List[Tuple[int, int]] = field(default_factory=list)      # (idx_i, idx_j)

    @property
    def stride(self) -> int:
        return self.S - self.ov

    @property
    def B(self) -> int:
        return len(self.origins)


def block_layout(variant: str, Ny: int, Nx: int, S: int = 128, ov: int | None = None) -> Layout:
    """Sliding-window enumeration of the three variants.

    chapter5: PM:306-329   (right->left, floor counts, extra ``-1`` column)
    deltas  : SMD:461-479  (right->left, ceil in x, tag = (i, n_x-j))
    gradp   : UGP:479-500  (left->right, ceil in x, tag = (i, j))
    """
    if ov is None:
        ov = default_overlap(variant, S)
    if Ny < S or Nx < S:
        raise ValueError("grid smaller than one block")
    st = S - ov
    lay = Layout(variant, Ny, Nx, S, ov, 0, 0)
    if variant == CHAPTER5:
        lay.n_x = int((Nx - S) / st)
        lay.n_y = int((Ny - S) / st)
        for i in range(lay.n_y + 2):
            y0 = Ny - S if i == lay.n_y + 1 else i * st
            for j in range(lay.n_x + 1):
                lay.origins.append((y0, Nx - S - j * st))
                lay.tags.append((i, lay.n_x - j))
                if j == lay.n_x:
                    lay.origins.append((y0, 0))
                    lay.tags.append((i, -1))
    elif variant == DELTAS:
        lay.n_x = int(math.ceil((Nx - S) / st))
        lay.n_y = int((Ny - S) / st)
        for i in range(lay.n_y + 2):
            y0 = Ny - S if i == lay.n_y + 1 else i * st
            for j in range(lay.n_x + 1):
                x0 = 0 if j == lay.n_x else Nx - S - j * st
                lay.origins.append((y0, x0))
                lay.tags.append((i, lay.n_x - j))
    elif variant == GRADP:
        lay.n_x = int(math.ceil((Nx - S) / st))
        lay.n_y = int((Ny - S) / st)
        for i in range(lay.n_y + 2):
            y0 = Ny - S if i == lay.n_y + 1 else i * st
            for j in range(lay.n_x + 1):
                x0 = Nx - S if j == lay.n_x else j * st
                lay.origins.append((y0, x0))
                lay.tags.append((i, j))
    else:
        raise ValueError(f"unknown variant {variant!r}")
    return lay


def extract_blocks(grid: np.ndarray, lay: Layout, c_in: int) -> np.ndarray:
    """``x_array`` of PM:332 / SMD:481 / UGP:502: [B, S, S, c_in] copies of the
    first ``c_in`` channels of ``grid[Ny, Nx, C]``."""
    S = lay.S
    out = np.empty((lay.B, S, S, c_in), dtype=grid.dtype)
    for b, (y0, x0) in enumerate(lay.origins):
        out[b] = grid[y0:y0 + S, x0:x0 + S, :c_in]
    return out


# --------------------------------------------------------------------------
# a9 / a10 / a11   PCA encode, MLP, PCA decode
# --------------------------------------------------------------------------
@dataclass
class Scaler:
    """PCA-coefficient scaling, UTL:290-329 / SMD:505-539 / PM:351,365."""
    kind: str = "max_abs"            # max_abs | std | min_max
    in_a: np.ndarray | float = 1.0   # max_abs_in | mean_in | min_in
    in_b: np.ndarray | float = 1.0   # (unused)   | std_in  | max_in
    out_a: np.ndarray | float = 1.0  # max_abs_out| mean_out| min_out
    out_b: np.ndarray | float = 1.0  # (unused)   | std_out | max_out

    def fwd(self, t):
        if self.kind == "max_abs":
            return t / self.in_a
        if self.kind == "std":
            return (t - self.in_a) / self.in_b
        if self.kind == "min_max":
            return (t - self.in_a) / (self.in_b - self.in_a)
        raise ValueError("Standardization method not valid")

    def inv(self, r):
        if self.kind == "max_abs":
            return r * self.out_a
        if self.kind == "std":
            return r * self.out_b + self.out_a
        if self.kind == "min_max":
            return r * (self.out_b - self.out_a) + self.out_a
        raise ValueError("Standardization method not valid")


@dataclass
class Model:
    """Everything the surrogate needs besides the grid."""
    variant: str
    c_in: int
    c_out: int
    comp_in: np.ndarray       # [P_i, S*S*c_in]   sklearn components_[:P_i]
    mean_in: np.ndarray       # [S*S*c_in]        sklearn mean_
    comp_out: np.ndarray      # [P_o, S*S*c_out]
    mean_out: np.ndarray      # [S*S*c_out]
    weights: Sequence[Tuple[np.ndarray, np.ndarray]]   # [(W[in,out] f32, b[out] f32)], last = linear head
    scaler: Scaler = field(default_factory=Scaler)
    out_scale: float = 1.0    # SMD:551 max_abs_p * U_max^2 (1 for gradp, UGP:537-538)
    S: int = 128
    ov: int | None = None
    sdf_ch: int = 2
    conv1d: Sequence[Tuple[np.ndarray, np.ndarray]] = ()   # conv1D_PCA head (NNS:75-124): [(K[k, c_in, c_out] f32, b[c_out] f32)]
    attention: dict | None = None     # densePCA_attention (NNS:40-72): MultiHeadAttention + LayerNormalization parameters

    def overlap(self) -> int:
        return default_overlap(self.variant, self.S) if self.ov is None else self.ov


def pca_encode(x_blocks: np.ndarray, comp_in: np.ndarray, mean_in: np.ndarray) -> np.ndarray:
    """PM:344-349 / sklearn ``transform`` (SMD:494, UGP:518):
    ``(X - mean_) @ components_[:P].T`` in float64."""
    flat = x_blocks.reshape(x_blocks.shape[0], -1).astype(np.float64)
    return np.dot(flat - mean_in, comp_in.T)


def conv1d_forward(x: np.ndarray, convs) -> np.ndarray:
    """The Conv1D part of ``conv1D_PCA`` (NNS:81-114): Input(shape=(PC_input, 1)), then per layer
    ``Conv1D(filters, kernel_size, activation='relu', padding='same')`` -- a cross-correlation over the coefficient index
    with (k - 1) // 2 zeros in front (TensorFlow's 'same' rule) -- and ``Flatten`` ([p, c] row-major).  float32."""
    h = np.asarray(x, np.float32)[:, :, None]
    B, Pn = h.shape[:2]
    for K, b in convs:
        K = np.asarray(K, np.float32)
        k, cin, cout = K.shape
        front = (k - 1) // 2
        hp = np.pad(h, ((0, 0), (front, k - 1 - front), (0, 0)))
        cols = np.concatenate([hp[:, t:t + Pn, :] for t in range(k)], axis=2)          # [B, P, (t, ci)]
        h = np.maximum(cols.reshape(B * Pn, k * cin) @ K.reshape(k * cin, cout) + np.asarray(b, np.float32), np.float32(0)).reshape(B, Pn, cout)
    return h.reshape(B, -1)


def multi_head_attention(q_in: np.ndarray, kv_in: np.ndarray, att: dict) -> np.ndarray:
    """``tf.keras.layers.MultiHeadAttention(num_heads, key_dim)(query, value)`` as Keras defines it (third-party, un-vendored;
    call site NNS:55 with query = value = x[:, None, :]): EinsumDense projections to [batch, seq, heads, dim] with bias,
    scores = (q / sqrt(dim)) . k over dim, softmax over the KEY axis, context = scores . v, EinsumDense output projection
    [heads, dim] -> d with bias.  float32.  q_in [B, T, d], kv_in [B, S, d] -> [B, T, d]."""
    f = lambda a: np.asarray(a, np.float32)
    q = np.einsum("btd,dhk->bthk", f(q_in), f(att["Wq"])) + f(att["bq"])
    k = np.einsum("bsd,dhk->bshk", f(kv_in), f(att["Wk"])) + f(att["bk"])
    v = np.einsum("bsd,dhk->bshk", f(kv_in), f(att["Wv"])) + f(att["bv"])
    scores = np.einsum("bthk,bshk->bhts", q * np.float32(1.0 / np.sqrt(q.shape[-1])), k)
    scores = np.exp(scores - scores.max(axis=-1, keepdims=True))
    scores = (scores / scores.sum(axis=-1, keepdims=True)).astype(np.float32)
    ctx = np.einsum("bhts,bshk->bthk", scores, v)
    return (np.einsum("bthk,hkd->btd", ctx, f(att["Wo"])) + f(att["bo"])).astype(np.float32)


def layer_normalization(x: np.ndarray, gamma, beta, eps: float = 1e-3) -> np.ndarray:
    """``tf.keras.layers.LayerNormalization()`` (NNS:56, 64; Keras defaults: axis -1, epsilon 1e-3, center, scale): moments over
    the last axis (biased variance), (x - mean) * rsqrt(var + eps) * gamma + beta, float32."""
    x = np.asarray(x, np.float32)
    mean = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    var = np.mean((x - mean) ** 2, axis=-1, keepdims=True, dtype=np.float32)
    return ((x - mean) / np.sqrt(var + np.float32(eps)) * np.asarray(gamma, np.float32) + np.asarray(beta, np.float32)).astype(np.float32)


def mlp_attention_forward(x: np.ndarray, weights, att: dict, rnd=None) -> np.ndarray:
    """``densePCA_attention`` (NNS:40-72) at inference (Dropout = identity): ``weights`` = its n_layers Dense layers and the
    head, ``att`` = the MultiHeadAttention and the LayerNormalizations.  ``rnd`` (None = the reference's float32 network):
    rounding of the operands entering a contraction, for the emulation of the library's bf16 mode -- which runs the
    attention block as ONE folded affine layer (its softmax over a single key is 1), so that is what is rounded."""
    f = lambda a: np.asarray(a, np.float32)
    if rnd is None:
        dense = lambda h, W, b: h @ f(W) + f(b)
    else:
        dense = lambda h, W, b: (rnd(h).astype(np.float64) @ rnd(f(W)).astype(np.float64)).astype(np.float32) + f(b)
    eps = att.get("eps", 1e-3)
    h = np.maximum(dense(f(x), *weights[0]), np.float32(0))                     # NNS:49
    if rnd is None:                                                               # the block as Keras runs it
        a = multi_head_attention(h[:, None, :], h[:, None, :], att)[:, 0, :]      # NNS:54-55, 57
    else:                                                                         # bf16 emulation: the folded affine layer
        HV = int(np.prod(np.shape(att["Wv"])[1:]))
        Wf = f(att["Wv"]).reshape(-1, HV).astype(np.float64) @ f(att["Wo"]).reshape(HV, -1).astype(np.float64)
        bfold = f(att["bv"]).reshape(-1).astype(np.float64) @ f(att["Wo"]).reshape(HV, -1).astype(np.float64) + f(att["bo"])
        a = dense(h, Wf.astype(np.float32), bfold.astype(np.float32))
    a = layer_normalization(a, *att["ln"][0], eps)                               # NNS:56
    for i in range(1, len(weights) - 1):                                          # NNS:60-64
        xi = np.maximum(dense(a, *weights[i]), np.float32(0))
        a = layer_normalization(xi + a, *att["ln"][i], eps)
    return dense(a, *weights[-1])                                                 # NNS:66


def mlp_forward(x: np.ndarray, weights, conv1d=(), attention=None) -> np.ndarray:
    """Keras ``Dense`` stack (PM:121-134, NNS:8-38): relu(x@W+b) for every layer
    but the last, which is linear.  float32 like Keras (floatx).  ``conv1d``: the Conv1D layers of the conv1D_PCA head
    in front of it (NNS:75-124)."""
    h = np.asarray(x, dtype=np.float32)
    if attention:
        return mlp_attention_forward(h, weights, attention)
    if len(conv1d):
        h = conv1d_forward(h, conv1d)
    n = len(weights)
    for li, (W, b) in enumerate(weights):
        h = h @ np.asarray(W, np.float32) + np.asarray(b, np.float32)
        if li != n - 1:
            h = np.maximum(h, np.float32(0))
    return h


def label_blocks(grid: np.ndarray, labels: np.ndarray, lay: "Layout", c_in: int, sdf_ch: int = 2) -> np.ndarray:
    """SMD:483-489 (UGP:509-511): the label image cut into the layout's blocks, each de-meaned over its flow cells
    (``y -= mean(y[x[..., sdf] != 0])``; a block without flow cells keeps its values).  -> [B, S, S, c_out] float64."""
    lab = np.asarray(labels, np.float64)
    lab = lab[..., None] if lab.ndim == 2 else lab
    xb = extract_blocks(np.asarray(grid, np.float64), lay, c_in)
    yb = extract_blocks(lab, lay, lab.shape[-1]).copy()
    for b in range(lay.B):
        m = xb[b, :, :, sdf_ch] != 0
        if m.any():
            for ch in range(yb.shape[-1]):
                yb[b, :, :, ch][m] -= np.mean(yb[b, :, :, ch][m])
    return yb


def compute_in_block_error(pred: np.ndarray, true: np.ndarray, flow_bool: np.ndarray):
    """``utils.compute_in_block_error`` (pressureSM_deltas/utils.py:210-243; call site SMD:553-557): over the flow cells of all
    blocks, norm = max(true) - min(true), NaN differences left out -> (mean(pred - true) / norm, mean((pred - true)^2) / norm^2),
    the two values the evaluator appends to ``pred_minus_true_block`` / ``pred_minus_true_squared_block``."""
    fb = np.broadcast_to(flow_bool, np.shape(true))
    t, p = np.asarray(true)[fb], np.asarray(pred)[fb]
    norm = np.max(t) - np.min(t)
    d = p - t
    d = d[~np.isnan(d)]
    return float(np.mean(d) / norm), float(np.mean(d ** 2) / norm ** 2)


def pca_decode(res: np.ndarray, comp_out: np.ndarray, mean_out: np.ndarray, S: int, c_out: int) -> np.ndarray:
    """PM:365-366 / SMD:541-542 / UGP:533-534: ``res @ components_[:P] + mean_``
    reshaped to [B, S, S, c_out] (channel last, interleaved)."""
    flat = np.dot(np.asarray(res, np.float64), comp_out) + mean_out
    return flat.reshape(res.shape[0], S, S, c_out)


# --------------------------------------------------------------------------
# a12  reassembly -- one function per reference variant
# --------------------------------------------------------------------------
@dataclass
class Assembly:
    field: np.ndarray            # [Ny, Nx] float64
    offsets: np.ndarray          # [B] correction subtracted from each block (NaN for skipped)
    shift: float                 # final global reference shift
    covered: np.ndarray          # [Ny, Nx] bool, cells written by a paste


def assemble_deltas(pred, x_blocks, lay: Layout, ref_bc: float = 0.0, sdf_ch: int = 2,
                    degenerate: str = "skip") -> Assembly:
    """SMD:182-365 (``assemble_prediction`` with ``apply_filter=False`` and
    ``apply_deltaU_change_wgt=False`` as called at SMD:573-575)."""
    S, ov, n_x, n_y, Ny, Nx = lay.S, lay.ov, lay.n_x, lay.n_y, lay.Ny, lay.Nx
    st = S - ov
    if n_x < 1:
        raise ValueError("deltas reassembly needs at least two block columns (SMD:237-240 reads the previous block)")
    pred = np.array(pred, dtype=np.float64, copy=True)
    p_i = Ny - (st * n_y + S)                      # SMD:213
    p_j = Nx - (st * n_x + S)                      # SMD:216
    lim = ov - p_j                                 # SMD:238
    if p_i == 0 and degenerate != "skip":
        raise ValueError("reference undefined: p_i == 0 raises a broadcast error at SMD:335")
    out = np.zeros((Ny, Nx))
    cov = np.zeros((Ny, Nx), bool)
    offs = np.full(lay.B, np.nan)
    up = np.zeros(n_x + 1)                          # SMD:210 BC_ups
    prev = None
    for b, (ti, tj) in enumerate(lay.tags):
        if ti == n_y + 1 and p_i == 0:
            continue
        m = x_blocks[b, :, :, sdf_ch] != 0
        cur = pred[b]

        def side(w):        # right strip of this block against the left strip of the previous one
            return mmean(cur[:, S - w:], m[:, S - w:]) - mmean(prev[:, :w], m[:, :w])

        if ti == 0:                                              # SMD:228-246
            if b == 0:
                c = mmean(cur[:, S - 1], m[:, S - 1]) - ref_bc
            else:
                c = side(ov)
            if tj == 0:
                c = side(lim)
            cur -= c
            up[tj] = mmean(cur[S - ov:, :], m[S - ov:, :])
        elif ti != n_y + 1:                                      # SMD:249-283
            if math.isnan(up[tj]):
                if tj == 0:
                    c = side(lim)
                elif tj == n_x:
                    c = mmean(cur[:ov, :], m[:ov, :]) - up[tj]
                else:
                    c = side(ov)
            else:
                c = mmean(cur[:ov, :], m[:ov, :]) - up[tj]
            cur -= c
            up[tj] = mmean(cur[S - ov:, :], m[S - ov:, :])
            if ti == n_y:
                up[tj] = mmean(cur[p_i:, :], m[p_i:, :])         # rows -(S-p_i):
        else:                                                    # SMD:286-328
            a0, a1 = S - p_i - ov, S - p_i
            if tj == n_x:
                c = mmean(cur[a0:a1, :], m[a0:a1, :]) - up[tj]
            else:
                n_up = int(np.count_nonzero(m[a0:a1, :]))
                if n_up / 128 ** 2 > 0.9:                        # SMD:307 (hard-wired 128)
                    c = side(lim) if tj == 0 else side(ov)
                else:
                    c = mmean(cur[:S - p_i, :], m[:S - p_i, :]) - up[tj]
            cur -= c
        offs[b] = c
        prev = cur
        # ---- paste (SMD:334-348); later blocks overwrite earlier ones
        jr = n_x - tj
        xs = slice(0, S) if tj == 0 else slice(Nx - S - jr * st, Nx - jr * st)
        if ti == n_y + 1:
            out[Ny - p_i:, xs] = cur[S - p_i:, :]
            cov[Ny - p_i:, xs] = True
        else:
            out[st * ti: st * ti + S, xs] = cur
            cov[st * ti: st * ti + S, xs] = True
    shift = float(np.mean(3.0 * out[:, -1] - out[:, -2]) / 3.0)   # SMD:350
    out -= shift
    return Assembly(out, offs, shift, cov)


def assemble_gradp(which: str, pred, x_blocks, lay: Layout, ref_bc: float = 0.0, sdf_ch: int = 2,
                   degenerate: str = "skip") -> Assembly:
    """UGP:255-369 for one output channel; ``which`` is 'dp_dx' or 'dp_dy'."""
    if which not in ("dp_dx", "dp_dy"):
        raise ValueError(which)
    S, ov, n_x, n_y, Ny, Nx = lay.S, lay.ov, lay.n_x, lay.n_y, lay.Ny, lay.Nx
    st = S - ov
    if n_x < 1:
        raise ValueError("gradp reassembly needs at least two block columns (UGP:307-310)")
    pred = np.array(pred, dtype=np.float64, copy=True)
    p_i = Ny - (S * (n_y + 1) - n_y * ov)          # UGP:277
    p_j = (Nx - S) - n_x * st                      # UGP:278
    lim = ov - p_j                                 # UGP:308
    out = np.zeros((Ny, Nx))
    cov = np.zeros((Ny, Nx), bool)
    offs = np.full(lay.B, np.nan)
    up = np.zeros(n_x + 1)
    prev = None
    skip_last = (p_i == 0 and degenerate == "skip")
    for b, (ti, tj) in enumerate(lay.tags):
        if ti == n_y + 1 and skip_last:
            continue
        m = x_blocks[b, :, :, sdf_ch] != 0
        cur = pred[b]

        def side(w):        # left strip of this block against the right strip of the previous one
            return mmean(cur[:, :w], m[:, :w]) - mmean(prev[:, S - w:], m[:, S - w:])

        if ti == 0:                                              # UGP:288-312
            if b == 0:
                if which == "dp_dx":
                    col = 0
                    while not m[:, col].any():
                        col += 1
                        if col >= S:
                            raise ValueError("first block has no flow cell")
                    c = mmean(cur[:, col], m[:, col]) - ref_bc
                else:
                    c = mmean(cur[1, :], m[1, :]) - ref_bc
            else:
                c = side(ov)
            if tj == n_x:
                c = side(lim)
            cur -= c
            up[tj] = mmean(cur[S - ov:, :], m[S - ov:, :])
        elif ti != n_y + 1:                                      # UGP:314-328
            if math.isnan(up[tj]):
                c = side(lim) if tj == n_x else side(ov)
            else:
                c = mmean(cur[:ov, :], m[:ov, :]) - up[tj]
            cur -= c
            up[tj] = mmean(cur[S - ov:, :], m[S - ov:, :])
            if ti == n_y:
                up[tj] = mmean(cur[p_i:, :], m[p_i:, :])
        else:                                                    # UGP:330-341
            if math.isnan(up[tj]):
                c = side(lim) if tj == n_x else side(ov)
            else:
                # python slice [-p_i-ov:-p_i]; empty when p_i == 0 (strict mode)
                a0, a1 = (S - p_i - ov, S - p_i) if p_i != 0 else (S - ov, 0)
                c = mmean(cur[a0:a1, :], m[a0:a1, :]) - up[tj]
            cur -= c
        offs[b] = c
        prev = cur
        # ---- paste (UGP:345-356)
        ys = slice(Ny - st, Ny) if ti == n_y + 1 else slice(ti * st, ti * st + S)
        src_rows = slice(ov, S) if ti == n_y + 1 else slice(0, S)
        if tj == n_x:
            out[ys, Nx - lim:] = cur[src_rows, S - lim:]
            cov[ys, Nx - lim:] = True
        else:
            out[ys, tj * st: tj * st + S] = cur[src_rows, :]
            cov[ys, tj * st: tj * st + S] = True
    if which == "dp_dx":
        shift = float(np.mean(3.0 * out[:, 0] - out[:, 1]) / 3.0)    # UGP:359
    else:
        shift = float(np.mean(3.0 * out[1, :] - out[2, :]) / 3.0)    # UGP:361
    out -= shift
    return Assembly(out, offs, shift, cov)


def assemble_chapter5(pred, x_blocks, lay: Layout, sdf_ch: int = 2) -> Assembly:
    """PM:373-472 (inline correction loop of ``py_func``)."""
    S, av, n_x, n_y, Ny, Nx = lay.S, lay.ov, lay.n_x, lay.n_y, lay.Ny, lay.Nx
    st = S - av
    pred = np.array(pred, dtype=np.float64, copy=True)
    p = Ny - (S * (n_y + 1) - n_y * av)            # PM:410
    p_j = (Nx - S) - n_x * S + n_x * av            # PM:397
    out = np.zeros((Ny, Nx))
    cov = np.zeros((Ny, Nx), bool)
    offs = np.full(lay.B, np.nan)
    up = np.zeros(n_x + 1)                          # BC_ups
    up_m1 = float("nan")                            # BC_up_  (column tagged -1)
    ant0 = float("nan")                             # BC_ant_0
    alter = 0.0                                     # BC_alter
    R = slice(S - av, S)                            # last `avance` rows / columns
    for b, (ti, tj) in enumerate(lay.tags):
        m = x_blocks[b, :, :, sdf_ch] != 0
        cur = pred[b]
        C = slice(p_j, p_j + av)
        if ti == 0:                                              # PM:388-405
            if tj == n_x:
                c = mmean(cur[:, R], m[:, R]) - 0.0
                cur -= c
                up[tj] = mmean(cur[R, R], m[R, R])
            elif tj == -1:
                c = mmean(cur[:, C], m[:, C]) - ant0
                cur -= c
                up_m1 = mmean(cur[R, C], m[R, C])
            else:
                c = mmean(cur[:, R], m[:, R]) - ant0
                cur -= c
                up[tj] = mmean(cur[R, :], m[R, :])
            ant0 = mmean(cur[:, :av], m[:, :av])
        elif ti == n_y + 1:                                      # PM:407-423
            T = slice(S - p - av, S - p)
            if tj == -1:
                c = mmean(cur[T, C], m[T, C]) - up_m1
            elif math.isnan(up[tj]):
                c = mmean(cur[:, R], m[:, R]) - alter
            else:
                c = mmean(cur[T, :], m[T, :]) - up[tj]
            cur -= c
        else:                                                    # PM:425-441
            if tj == -1:
                c = mmean(cur[:av, C], m[:av, C]) - up_m1
                cur -= c
                up_m1 = float(np.mean(cur[R, C]))                # PM:432, unmasked
            else:
                if math.isnan(up[tj]):
                    c = mmean(cur[:, R], m[:, R]) - alter
                else:
                    c = mmean(cur[:av, :], m[:av, :]) - up[tj]
                cur -= c
                up[tj] = mmean(cur[R, :], m[R, :])
        alter = mmean(cur[:, :av], m[:, :av])                    # PM:445
        offs[b] = c
        # ---- paste (PM:449-467)
        if ti == n_y + 1 and tj == -1:
            w = Nx - (n_x + 1) * st - av
            out[Ny - st:, :w] = cur[av:, :w]
            cov[Ny - st:, :w] = True
        elif tj == -1:
            out[ti * st: ti * st + S, :S] = cur
            cov[ti * st: ti * st + S, :S] = True
        else:
            j = n_x - tj
            xs = slice(Nx - S - j * st, Nx - j * st)
            if ti == n_y + 1:
                out[Ny - st:, xs] = cur[av:, :]
                cov[Ny - st:, xs] = True
            else:
                out[ti * st: ti * st + S, xs] = cur
                cov[ti * st: ti * st + S, xs] = True
    shift = float(np.mean(3.0 * out[:, -1] - out[:, -2]) / 3.0)   # PM:472
    out -= shift
    return Assembly(out, offs, shift, cov)


# --------------------------------------------------------------------------
# whole grid-native solve  (PM:299-473, SMD:452-575, UGP:470-547)
# --------------------------------------------------------------------------
@dataclass
class Solve:
    x_blocks: np.ndarray      # [B,S,S,c_in]
    coeff_in: np.ndarray      # [B,P_i]  PCA coefficients before scaling (f64)
    x_input: np.ndarray       # [B,P_i]  scaled network input (f64)
    res: np.ndarray           # [B,P_o]  raw network output (f32)
    block_pred: np.ndarray    # [B,S,S,c_out] decoded (and out_scale'd) blocks (f64)
    fields: np.ndarray        # [Ny,Nx,c_out] assembled (f64)
    assemblies: list


def bf16_round(x) -> np.ndarray:
    """Round-to-nearest-even to bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    out[np.isnan(f)] = np.nan
    return out


def solve_grid(grid: np.ndarray, model: Model, degenerate: str = "skip", precision: str = "f32") -> Solve:
    """One grid-native solve.  ``grid`` is the already normalised
    [Ny, Nx, >=c_in] image of PM:288-297 / SMD:430-444 / UGP:453-468.

    ``precision='bf16'`` emulates the bf16 operand path of the HIP library (BASELINE config 4;
    no reference counterpart): bases, weights and the activations entering each contraction are
    rounded to bfloat16, products and sums are exact/float64 here (float32 on the GPU)."""
    Ny, Nx = grid.shape[:2]
    lay = block_layout(model.variant, Ny, Nx, model.S, model.overlap())
    xb = extract_blocks(np.asarray(grid, np.float64), lay, model.c_in)
    if precision == "bf16" and len(model.conv1d):
        raise ValueError("the conv1D_PCA head has no bf16 path")
    if precision == "bf16":
        flat = xb.reshape(lay.B, -1).astype(np.float32) - model.mean_in.astype(np.float32)
        coeff = bf16_round(flat).astype(np.float64) @ bf16_round(model.comp_in).astype(np.float64).T
        x_in = model.scaler.fwd(coeff)
        h = np.asarray(x_in, np.float32)
        if model.attention:
            h = mlp_attention_forward(h, model.weights, model.attention, rnd=bf16_round)
        else:
            for li, (W, b) in enumerate(model.weights):
                h = (bf16_round(h).astype(np.float64) @ bf16_round(W).astype(np.float64)).astype(np.float32) + np.asarray(b, np.float32)
                if li != len(model.weights) - 1:
                    h = np.maximum(h, np.float32(0))
        res = h
        dec_in = model.scaler.inv(res.astype(np.float64))
        flat_out = bf16_round(dec_in).astype(np.float64) @ bf16_round(model.comp_out).astype(np.float64) + model.mean_out
        bp = flat_out.reshape(lay.B, model.S, model.S, model.c_out) * model.out_scale
    elif precision == "f32":
        coeff = pca_encode(xb, model.comp_in, model.mean_in)
        x_in = model.scaler.fwd(coeff)
        res = mlp_forward(x_in, model.weights, model.conv1d, model.attention)
        dec_in = model.scaler.inv(res.astype(np.float64))
        bp = pca_decode(dec_in, model.comp_out, model.mean_out, model.S, model.c_out) * model.out_scale
    else:
        raise ValueError(precision)
    fields = np.zeros((Ny, Nx, model.c_out))
    asm = []
    if model.variant == CHAPTER5:
        a = assemble_chapter5(bp[..., 0], xb, lay, model.sdf_ch)
        fields[..., 0] = a.field
        asm.append(a)
    elif model.variant == DELTAS:
        a = assemble_deltas(bp[..., 0], xb, lay, 0.0, model.sdf_ch, degenerate)
        fields[..., 0] = a.field
        asm.append(a)
    else:
        for ch, which in enumerate(("dp_dx", "dp_dy")):
            a = assemble_gradp(which, bp[..., ch], xb, lay, 0.0, model.sdf_ch, degenerate)
            fields[..., ch] = a.field
            asm.append(a)
    return Solve(xb, coeff, x_in, res, bp, fields, asm)


# --------------------------------------------------------------------------
# a1 / a5  grid + barycentric interpolation (mesh <-> grid, "next" rows)
# --------------------------------------------------------------------------
def create_uniform_grid(x_min, x_max, y_min, y_max, delta):
    """PM:42-48 / UTL:111-125: cell-centred uniform grid, flattened row-major
    with y as the outer index."""
    nx = int(round((x_max - x_min) / delta))
    ny = int(round((y_max - y_min) / delta))
    X0 = np.linspace(x_min + delta / 2, x_max - delta / 2, num=nx)
    Y0 = np.linspace(y_min + delta / 2, y_max - delta / 2, num=ny)
    XX, YY = np.meshgrid(X0, Y0)
    return XX.flatten(), YY.flatten()


def interpolate(values, vtx, wts):
    """PM:64-65: ``sum_j values[vtx[n,j]] * wts[n,j]``."""
    return np.einsum("nj,nj->n", np.take(values, vtx), wts)


def interpolate_fill(values, vtx, wts, fill_value=np.nan):
    """PM:67-70 / UTL:75-90: as ``interpolate`` with ``fill_value`` wherever a
    barycentric weight is negative (target outside the triangulation)."""
    ret = interpolate(values, vtx, wts)
    ret[np.any(wts < 0, axis=1)] = fill_value
    return ret


# --------------------------------------------------------------------------
# mesh-side ends of the solver call (a1-a6, a14): init_func / py_func on one rank
# --------------------------------------------------------------------------
@dataclass
class Geometry:
    ny: int
    nx: int
    vert_m2g: np.ndarray     # vert_OFtoNP   (PM:210)
    wts_m2g: np.ndarray      # weights_OFtoNP
    vert_g2m: np.ndarray     # vert_NPtoOF   (PM:211)
    wts_g2m: np.ndarray
    indices: np.ndarray      # [ny*nx, 2] (ii, jj)  (PM:225-243)
    sdfunct: np.ndarray      # [ny, nx]


def interp_weights(xyz, uvw):
    """PM:52-62 (qhull Delaunay of the source points, barycentric coordinates of the targets)."""
    from scipy.spatial import Delaunay
    tri = Delaunay(xyz)
    simplex = tri.find_simplex(uvw)
    vertices = np.take(tri.simplices, simplex, axis=0)
    T = np.take(tri.transform, simplex, axis=0)
    bary = np.einsum("njk,nk->nj", T[:, :2, :], uvw - T[:, 2])
    return vertices, np.hstack((bary, 1 - bary.sum(axis=1, keepdims=True)))


def _inside_convex_hull(pts, cloud):
    """PM:83-90: shapely ``MultiPoint(cloud).convex_hull`` (un-vendored; its exterior ring is GEOS's clockwise
    closed ring) + matplotlib ``Path.contains_points`` (un-vendored; published algorithm: the crossings test of
    matplotlib/src/_path.h ``point_in_path_impl``, radius 0).  Scalar restatement, one point at a time."""
    from scipy.spatial import ConvexHull
    ring = [tuple(cloud[i]) for i in ConvexHull(cloud).vertices][::-1]     # qhull: counter-clockwise -> clockwise
    ring.append(ring[0])
    out = np.zeros(len(pts), bool)
    for i, (tx, ty) in enumerate(pts):
        if not (np.isfinite(tx) and np.isfinite(ty)):
            continue
        inside = False
        for (x0, y0), (x1, y1) in zip(ring, ring[1:] + ring[:1]):
            f0, f1 = y0 >= ty, y1 >= ty
            if f0 != f1 and (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1):
                inside = not inside
        out[i] = inside
    return out


def interp_weights_idw(xyz, uvw):
    """pressureSM_deltas/utils.py:22-55 (= pressureSM_Poisson/SM_call.py:139-172): ``interp_weights`` whose targets
    outside the hull take the 3 nearest source points with weights 1/max(d^2, 1e-6), normalised (sklearn KDTree in
    the reference; the neighbour set is unique up to exact distance ties)."""
    vertices, wts = interp_weights(xyz, uvw)
    from scipy.spatial import Delaunay
    outside = np.flatnonzero(Delaunay(xyz).find_simplex(uvw) == -1)
    vertices, wts = vertices.copy(), wts.copy()
    for t in outside:
        d2 = ((np.asarray(xyz) - np.asarray(uvw)[t]) ** 2).sum(axis=1)
        nn = np.argsort(d2, kind="stable")[:3]
        w = 1.0 / np.maximum(np.sqrt(d2[nn]) ** 2, 1e-6)
        vertices[t] = nn
        wts[t] = w / w.sum()
    return vertices, wts


def domain_dist(top, obst, xy0, every=10):
    """PM:72-99."""
    from scipy.spatial.distance import cdist
    box = ((xy0[:, 0] <= top[:, 0].max()) & (xy0[:, 0] >= top[:, 0].min())
           & (xy0[:, 1] <= top[:, 1].max()) & (xy0[:, 1] >= top[:, 1].min()))
    dom = box & ~_inside_convex_hull(xy0, obst)
    d = np.minimum(cdist(xy0, obst[::every]).min(axis=1), cdist(xy0, top[::every]).min(axis=1))
    return dom, d * dom


def init_geometry(array, top, obst, delta=5e-3, every=10) -> Geometry:
    """``init_func`` on one rank, PM:195-243, with ``indices`` zero-initialised (SMD:161; PM:225
    leaves it uninitialised)."""
    x_min, x_max = round(np.min(array[:, 2]), 2), round(np.max(array[:, 2]), 2)
    y_min, y_max = round(np.min(array[:, 3]), 2), round(np.max(array[:, 3]), 2)
    X0, Y0 = create_uniform_grid(x_min, x_max, y_min, y_max, delta)
    xy0 = np.c_[X0, Y0]
    pts = array[:, 2:4]
    v1, w1 = interp_weights(pts, xy0)
    v2, w2 = interp_weights(xy0, pts)
    dom, sdf = domain_dist(top, obst, xy0, every)
    ny, nx = int(round((y_max - y_min) / delta)), int(round((x_max - x_min) / delta))
    x0, y0 = X0.min(), Y0.min()
    ux = interpolate_fill(array[:, 0], v1, w1)
    indices = np.zeros((len(X0), 2), int)
    sdfunct = np.zeros((ny, nx))
    for step in range(len(X0)):
        if dom[step] and not np.isnan(ux[step]):
            jj = int(round((X0[step] - x0) / delta))
            ii = int(round((Y0[step] - y0) / delta))
            indices[step] = (ii, jj)
            sdfunct[ii, jj] = sdf[step]
    return Geometry(ny, nx, v1, w1, v2, w2, indices, sdfunct)


def mesh_to_grid(array, geo: Geometry, max_abs_Ux, max_abs_Uy, sdf_div=1.0, fill=False):
    """PM:267-297 (``fill=False``) / SMD:404-444 (``fill=True``, ``sdf_div=max_abs_dist``)."""
    U_max = np.max(np.sqrt(np.square(array[:, 0]) + np.square(array[:, 1])))
    f = interpolate_fill if fill else interpolate
    ux = f(array[:, 0] / U_max, geo.vert_m2g, geo.wts_m2g)
    uy = f(array[:, 1] / U_max, geo.vert_m2g, geo.wts_m2g)
    grid = np.zeros((geo.ny, geo.nx, 3))
    idx = tuple(geo.indices.T)
    grid[:, :, 0][idx] = ux / max_abs_Ux
    grid[:, :, 1][idx] = uy / max_abs_Uy
    grid[:, :, 2] = geo.sdfunct / sdf_div
    grid[np.isnan(grid)] = 0
    return grid, U_max


def grid_to_mesh(result, array, geo: Geometry, max_abs_p, U_max, wall=0.05):
    """PM:481-496."""
    p_unif = result[tuple(geo.indices.T)]
    p_interp = interpolate_fill(p_unif, geo.vert_g2m, geo.wts_g2m)
    p = p_interp * max_abs_p * U_max ** 2
    sdf_mesh = interpolate_fill(geo.sdfunct, geo.vert_g2m, geo.wts_g2m)
    prev = array[:, 4]
    with np.errstate(invalid="ignore"):
        near = sdf_mesh < wall
    p[near] = prev[near]
    nanm = np.isnan(p_interp)
    p[nanm] = prev[nanm]
    return p


def py_func_mesh(array, geo: Geometry, model: Model, maxs):
    """One serial ``py_func`` call (PM1:199-444 / PM:249-517 on one rank): cells[N,5] -> p[N]."""
    grid, U_max = mesh_to_grid(array, geo, maxs[0], maxs[1])
    sol = solve_grid(grid, model)
    return grid_to_mesh(sol.fields[..., 0], array, geo, maxs[3], U_max), grid, sol


# --------------------------------------------------------------------------
# U_to_gradP: integration of the assembled gradient into p (UGP:371-416, 592-628)
# --------------------------------------------------------------------------
def integrate_field(block, sdfunct, dx, dy, direction_x=1, direction_y=1):
    """UGP:371-416.  ``block`` [h,w,2] = (dp/dx, dp/dy).  Quirks kept: row ``i`` of the GLOBAL
    ``sdfunct`` (full width, cast to int) is used as an index list into the block row ("reset the
    cumulative sum at the obstacle"), whatever the block's position."""
    gx = np.array(block[..., 0], dtype=np.float64)
    gy = np.array(block[..., 1], dtype=np.float64)
    h, w = gx.shape
    SdPx = np.empty((h, w))
    for i in range(h):
        aaa = gx[i].copy()
        ccc = np.cumsum(aaa)
        nn = sdfunct[i, :].astype(int)
        dd = np.diff(np.concatenate(([0.0], ccc[nn])))
        aaa[nn] = -dd
        SdPx[i] = np.cumsum(aaa) * dx
    SdPy = np.cumsum(gy, axis=0) * dy
    ij = -1 if direction_x == -1 else 0
    ii = -1 if direction_y == -1 else 0
    return SdPy[:, ij][:, None] - SdPy[ii, ij] + SdPx - SdPx[:, ij][:, None]


def integrate_gradp(gradP, sdfunct, dx, dy, center_y, center_x):
    """UGP:597-628: the domain is cut into four quadrants at (center_y, center_x); each is integrated
    from its outer corner and the left quadrants are shifted onto the right ones over the flow cells
    of the two columns at the cut."""
    Ny, Nx = gradP.shape[:2]
    cy, cx = center_y, center_x
    out = np.empty((Ny, Nx))
    p1 = integrate_field(gradP[:cy, cx - 1:, :], sdfunct, dx, dy, direction_x=-1)
    out[:cy, cx - 1:] = p1
    p2 = integrate_field(gradP[:cy, :cx, :], sdfunct, dx, dy)
    m1, m2 = sdfunct[:cy, cx - 1] != 0, sdfunct[:cy, cx] != 0
    out[:cy, :cx] = p2 - (p2[:, -1][m2] - p1[:, 0][m1]).mean()
    p3 = integrate_field(gradP[cy:, cx - 1:, :], sdfunct, dx, dy, direction_x=-1, direction_y=-1)
    out[cy:, cx - 1:] = p3
    p4 = integrate_field(gradP[cy:, :cx, :], sdfunct, dx, dy, direction_y=-1)
    m3, m4 = sdfunct[cy:, cx - 1] != 0, sdfunct[cy:, cx] != 0
    out[cy:, :cx] = p4 - (p4[:, -1][m4] - p3[:, 0][m3]).mean()
    return out


def integration_center(sdfunct, x_min, x_max, X0_min, delta, row=200):
    """UGP:594-600: ``center_p_x`` from the obstacle's extent on grid row ``row`` (hard-wired 200)."""
    xl = np.linspace(x_min, x_max, sdfunct.shape[1])
    solid = sdfunct[row, :] == 0
    return int(((xl[solid].max() + xl[solid].min()) / 2 - X0_min) / delta), int(row)


# --------------------------------------------------------------------------------------
# pressureSM_Poisson input features (Improved_SM/.../pressureSM_Poisson/SM_call.py = SMP)
# --------------------------------------------------------------------------------------
def arcsinh_smooth_transform(field, k):
    """SMP:22-69 (`smart_arcsin_smooth_transform`, scale_output=False): central range
    [mean-k*std, mean+k*std] of the WHOLE field mapped to [-1,1], linear tails, then arcsinh."""
    field = np.asarray(field, np.float64)
    mean, std = np.mean(field), np.std(field)
    lo, hi = mean - k * std, mean + k * std
    with np.errstate(all="ignore"):
        below = -1.0 - (field - lo) / lo                    # SMP:50
        above = 1.0 + (field - hi) / hi                     # SMP:52
        mid = 2.0 * (field - lo) / (hi - lo) - 1.0          # SMP:54
    scaled = np.where(field < lo, below, np.where(field > hi, above, mid))
    return np.arcsinh(scaled)                               # SMP:60


def masked_gradient(grid):
    """SMP:602-621 followed by SMP:629-632: `np.gradient` (unit spacing), zero wherever the cell
    or one of its four direct neighbours is NaN.  Returns (d/dy, d/dx)."""
    g = np.asarray(grid, np.float64)
    nan = np.isnan(g)
    bad = nan.copy()
    bad[1:, :] |= nan[:-1, :]
    bad[:-1, :] |= nan[1:, :]
    bad[:, 1:] |= nan[:, :-1]
    bad[:, :-1] |= nan[:, 1:]
    with np.errstate(all="ignore"):
        gy, gx = np.gradient(g)
    gy[bad] = 0.0
    gx[bad] = 0.0
    return gy, gx


def poisson_features(ux, uy, dux, duy, sdfunct, L, U, k, max_abs):
    """SMP:588-711: grid image [Ny,Nx,4] fed to the deltas layout with C_in = 4.
    ux, uy, dux, duy: interpolated (dimensional) grids, zero outside the flow; sdfunct: raw
    signed-distance image (0 inside solids); max_abs = (Poisson_term_1, delta_Ux, delta_Uy, dist)."""
    ux = np.array(ux, np.float64); uy = np.array(uy, np.float64)
    solid = np.asarray(sdfunct) == 0
    ux[solid] = np.nan                                       # SMP:624-625
    uy[solid] = np.nan
    dUx_dy, dUx_dx = masked_gradient(ux)
    dUy_dy, dUy_dx = masked_gradient(uy)
    term = (dUx_dx * dUx_dx + 2 * dUx_dy * dUy_dx + dUy_dy * dUy_dy) * L ** 2 / U ** 2    # SMP:635
    grid = np.zeros(ux.shape + (4,))
    grid[..., 0] = arcsinh_smooth_transform(term, k)         # SMP:646, 698
    grid[..., 1] = np.asarray(dux, np.float64) / U           # SMP:641, 699
    grid[..., 2] = np.asarray(duy, np.float64) / U
    grid[..., 3] = np.asarray(sdfunct, np.float64)
    grid[np.isnan(grid)] = 0                                 # SMP:704
    grid /= np.asarray(max_abs, np.float64)                  # SMP:707-710
    return grid, term


# --------------------------------------------------------------------------------------
# dataset-driven evaluator front end (pressureSM_deltas/SM_call.py:381-451)
# --------------------------------------------------------------------------------------
def evaluator_grid_deltas(cells, vert, weights, indices, sdfunct, maxs):
    """SMD:381-451: one un-padded dataset frame cells[N,11] (0 Ux, 1 Uy, 2 p, 3 Cx, 4 Cy, 5-6 delta_U,
    7 delta_p, 8-9 delta_U_prev, 10 delta_p_prev) -> (grid[Ny,Nx,5] normalised, deltaU_change_grid,
    deltaP_prev_grid, U_max_norm), or None for an irrelevant time step (SMD:413-421)."""
    d = np.asarray(cells)        # dtype kept: the dataset is float32 and the reference normalises in float32
    Ux, Uy, p = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    delta_U, delta_p = d[:, 5:7], d[:, 7:8]
    delta_U_prev, delta_p_prev = d[:, 8:10], d[:, 10:11]
    changed = np.abs(delta_U - delta_U_prev).sum(axis=-1)
    changed = changed / changed.max()
    U = np.max(np.sqrt(np.square(Ux) + np.square(Uy)))
    dU = np.max(np.sqrt(np.square(delta_U[:, 0:1]) + np.square(delta_U[:, 1:2])))
    if (dU / U) < 1e-4:
        return None
    ny, nx = sdfunct.shape[:2]
    idx = tuple(np.asarray(indices).T)

    def to_grid(v):
        g = np.zeros((ny, nx))
        g[idx] = interpolate_fill(np.asarray(v).reshape(-1), vert, weights)     # NumPy order: last write wins
        return g
    grid = np.zeros((ny, nx, 5))
    grid[..., 0] = to_grid(delta_U[:, 0] / U)
    grid[..., 1] = to_grid(delta_U[:, 1] / U)
    grid[..., 2] = np.asarray(sdfunct).reshape(ny, nx)
    grid[..., 3] = to_grid(delta_p / pow(U, 2.0))
    grid[..., 4] = to_grid(p)
    grid[np.isnan(grid)] = 0
    grid[..., 0] /= maxs[0]; grid[..., 1] /= maxs[1]; grid[..., 2] /= maxs[2]; grid[..., 3] /= maxs[3]
    return grid, to_grid(changed), to_grid(delta_p_prev), float(U)


def evaluator_grid_gradp(cells, vert, weights, indices, sdfunct, maxs, min_x, max_x, min_y, max_y):
    """UGP:429-467: one un-padded dataset frame (0 Ux, 1 Uy, 2 p, 3 Cx, 4 Cy, 6 dP/dx, 7 dP/dy) ->
    (grid[Ny,Nx,6] normalised, U_max_norm).  dtype of ``cells`` and of the extents is kept (float32 in
    the reference: file data and ``np.max(top[:,0])``)."""
    d = np.asarray(cells)
    Ux, Uy, p, dPdx, dPdy = d[:, 0:1], d[:, 1:2], d[:, 2:3], d[:, 6:7], d[:, 7:8]
    U = np.max(np.sqrt(np.square(Ux) + np.square(Uy)))
    dPdx_adim = dPdx * (max_x - min_x) / pow(U, 2.0)
    dPdy_adim = dPdy * (max_y - min_y) / pow(U, 2.0)
    p_adim = p / pow(U, 2.0)
    ny, nx = sdfunct.shape[:2]
    idx = tuple(np.asarray(indices).T)

    def to_grid(v):
        g = np.zeros((ny, nx))
        g[idx] = interpolate_fill(np.asarray(v).reshape(-1), vert, weights)
        return g
    grid = np.zeros((ny, nx, 6))
    grid[..., 0] = to_grid(Ux / U)
    grid[..., 1] = to_grid(Uy / U)
    grid[..., 2] = np.asarray(sdfunct).reshape(ny, nx)
    grid[..., 3] = to_grid(dPdx_adim)
    grid[..., 4] = to_grid(dPdy_adim)
    grid[..., 5] = to_grid(p_adim)
    grid[np.isnan(grid)] = 0
    for ch in range(5):
        grid[..., ch] /= maxs[ch]
    return grid, float(U)
