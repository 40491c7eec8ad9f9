#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN
STATEMENTS on seeded synthetic inputs.  Runs only in the build container (it
needs /root/reference); the GPU box and the test-suite only read the ``.npz``
files it writes.

How the reference is executed although its modules cannot be imported here
(every one imports TensorFlow / h5py / mpi4py / shapely at module level and
those are not installed): the reference ``.py`` files are parsed
with ``ast`` *at run time*, the statements of the hot path (block extraction,
PCA encode, scaling, PCA decode, ``assemble_prediction`` / the inline
correction loop, global shift) are compiled unmodified and executed with

  * ``np``            -> the real NumPy,
  * ``self.pcainput`` -> a real scikit-learn ``PCA`` object carrying the seeded
                         synthetic ``components_`` / ``mean_`` (the reference's
                         pickles are missing from the snapshot),
  * ``self.model`` / ``model`` -> a NumPy callable ``relu(x@W+b)`` stack with
                         Keras ``Dense`` semantics (TensorFlow is third-party,
                         un-vendored and not installable here).

Statements that only print, plot or compute error metrics are dropped.  No
reference source text is stored in this repository: only inputs' seeds and the
numeric outputs.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import ast
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import psm_amd  # noqa: E402
from psm_amd import formats, synthetic  # noqa: E402

REF = "/root/reference"
PM = f"{REF}/Thesis_Work/Chapter5/parallelized/test_case/python_module.py"
SMD = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/SM_call.py"
UGP = f"{REF}/Improved_SM/U_to_gradP/evaluation/Eval_dual_Dense_onlycil.py"
WEIGHTS = f"{REF}/Thesis_Work/Chapter5/parallelized/test_case/weights.h5"
MAXS = f"{REF}/Thesis_Work/Chapter5/parallelized/test_case/maxs"
MAXS_PCA = f"{REF}/Thesis_Work/Chapter5/parallelized/test_case/maxs_PCA"

DROP = ("print(", "utils.", "plt.", "pred_minus_true", "fig.", "axs[")


# --------------------------------------------------------------------------
# ast helpers
# --------------------------------------------------------------------------
def _tree(path):
    with open(path) as f:
        return ast.parse(f.read(), filename=path)


def _find_fn(tree, name, cls=None):
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    return next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)


def _src(node):
    return ast.unparse(node)


def _slice(stmts, first_pred, last_pred):
    i0 = next(i for i, s in enumerate(stmts) if first_pred(_src(s)))
    i1 = next(i for i, s in enumerate(stmts) if i >= i0 and last_pred(_src(s)))
    keep = [s for s in stmts[i0:i1 + 1] if not any(d in _src(s) for d in DROP)]
    return keep


def _run(stmts, glb, loc, filename):
    mod = ast.Module(body=stmts, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, filename, "exec"), glb, loc)


def _method(tree, cls, name, glb, filename):
    fn = _find_fn(tree, name, cls)
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(glb)
    exec(compile(mod, filename, "exec"), ns)
    return ns[name]


# --------------------------------------------------------------------------
# stand-ins for the missing artefacts (NOT for arithmetic of the reference)
# --------------------------------------------------------------------------
def sk_pca(comp, mean):
    from sklearn.decomposition import PCA
    p = PCA(n_components=comp.shape[0])
    p.components_ = np.asarray(comp, np.float64)
    p.mean_ = np.asarray(mean, np.float64)
    p.n_features_in_ = comp.shape[1]
    p.n_components_ = comp.shape[0]
    p.whiten = False
    ev = np.linspace(2.0, 1.0, comp.shape[0])
    p.explained_variance_ = ev
    p.explained_variance_ratio_ = ev / ev.sum()
    return p


def dense_callable(weights):
    def model(x):
        h = np.asarray(x, np.float32)
        for li, (W, b) in enumerate(weights):
            h = h @ W + b
            if li != len(weights) - 1:
                h = np.maximum(h, np.float32(0))
        return h
    return model


class _KerasStub:
    """Stand-ins for the Keras symbols the reference's ``conv1D_PCA`` (NNs.py:75-124) touches, so that the FUNCTION
    ITSELF can be executed: it decides which layers exist, in which order, with which kernel size, padding and activation,
    where the Flatten sits and what the head is.  The arithmetic of the layers is restated in NumPy float32 (TensorFlow is
    not installable here): Conv1D as an explicit sum over taps of cross-correlations with TensorFlow's 'same' padding,
    Dense as x @ W + b.  Weights are drawn in layer-creation order from one seeded generator."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.layers = []
        stub = self

        class Sym:
            def __init__(self, fn, shape=None):
                self.fn, self.shape = fn, shape

            def __add__(self, other):                 # `x + attn_output` (NNs.py:64): element-wise sum of two symbolic tensors
                return Sym(lambda v: self.fn(v) + other.fn(v))

        def Input(shape=None, **kw):
            shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)

            def feed(x):
                x = np.asarray(x, np.float32)
                return x[..., None] if x.ndim == len(shape) else x      # functional models expand a missing last axis of 1
            return Sym(feed, shape)

        class Layer:
            def __call__(self, x, *more):
                if more:                              # MultiHeadAttention(query, value)
                    return Sym(lambda v: self.apply(x.fn(v), *[m.fn(v) for m in more]))
                return Sym(lambda v: self.apply(x.fn(v)))

        class Conv1D(Layer):
            def __init__(self, filters, kernel_size, activation=None, padding="valid", kernel_regularizer=None, **kw):
                self.filters, self.k, self.activation, self.padding = int(filters), int(kernel_size), activation, padding
                self.K = self.b = None
                stub.layers.append(self)

            def apply(self, h):
                if self.K is None:
                    cin = h.shape[2]
                    self.K = (stub.rng.standard_normal((self.k, cin, self.filters)) * np.sqrt(2.0 / (self.k * cin))).astype(np.float32)
                    self.b = (stub.rng.standard_normal(self.filters) * 0.05).astype(np.float32)
                assert self.padding == "same"
                n = h.shape[1]
                front = (self.k - 1) // 2
                hp = np.pad(h, ((0, 0), (front, self.k - 1 - front), (0, 0)))
                out = np.zeros((h.shape[0], n, self.filters), np.float32) + self.b
                for t in range(self.k):
                    out = out + hp[:, t:t + n, :] @ self.K[t]
                return np.maximum(out, np.float32(0)) if self.activation == "relu" else out

        class Dense(Layer):
            def __init__(self, units, activation=None, kernel_regularizer=None, **kw):
                self.units, self.activation, self.W = int(units), activation, None
                stub.layers.append(self)

            def apply(self, h):
                if self.W is None:
                    self.W = (stub.rng.standard_normal((h.shape[1], self.units)) * np.sqrt(1.0 / h.shape[1])).astype(np.float32)
                    self.b = (stub.rng.standard_normal(self.units) * 0.05).astype(np.float32)
                out = h @ self.W + self.b
                return np.maximum(out, np.float32(0)) if self.activation == "relu" else out

        class MultiHeadAttention(Layer):
            """Keras' published definition restated in NumPy float32 (query / key / value EinsumDense projections with bias,
            scaled dot product, softmax over the keys, EinsumDense output projection), for whatever sequence lengths the
            reference's call produces."""
            def __init__(self, num_heads, key_dim, **kw):
                self.h, self.kd, self.P = int(num_heads), int(key_dim), None
                stub.layers.append(self)

            def apply(self, q_in, v_in):
                d = q_in.shape[-1]
                if self.P is None:
                    g = lambda *sh: (stub.rng.standard_normal(sh) * np.sqrt(1.0 / sh[0])).astype(np.float32)
                    bias = lambda *sh: (stub.rng.standard_normal(sh) * 0.05).astype(np.float32)
                    self.P = dict(Wq=g(d, self.h, self.kd), bq=bias(self.h, self.kd), Wk=g(d, self.h, self.kd), bk=bias(self.h, self.kd),
                                  Wv=g(d, self.h, self.kd), bv=bias(self.h, self.kd),
                                  Wo=(stub.rng.standard_normal((self.h, self.kd, d)) * np.sqrt(1.0 / (self.h * self.kd))).astype(np.float32),
                                  bo=bias(d))
                P = self.P
                q = np.einsum("btd,dhk->bthk", q_in, P["Wq"]) + P["bq"]
                k = np.einsum("bsd,dhk->bshk", v_in, P["Wk"]) + P["bk"]
                v = np.einsum("bsd,dhk->bshk", v_in, P["Wv"]) + P["bv"]
                sc = np.einsum("bthk,bshk->bhts", q * np.float32(1.0 / np.sqrt(self.kd)), k)
                sc = np.exp(sc - sc.max(axis=-1, keepdims=True))
                sc = (sc / sc.sum(axis=-1, keepdims=True)).astype(np.float32)
                ctx = np.einsum("bhts,bshk->bthk", sc, v)
                return (np.einsum("bthk,hkd->btd", ctx, P["Wo"]) + P["bo"]).astype(np.float32)

        class LayerNormalization(Layer):
            """Keras defaults: axis -1, epsilon 1e-3, centre and scale; gamma / beta drawn away from (1, 0) as trained ones are."""
            def __init__(self, axis=-1, epsilon=1e-3, **kw):
                assert axis == -1
                self.eps, self.gamma = float(epsilon), None
                stub.layers.append(self)

            def apply(self, h):
                if self.gamma is None:
                    self.gamma = (1.0 + 0.1 * stub.rng.standard_normal(h.shape[-1])).astype(np.float32)
                    self.beta = (0.05 * stub.rng.standard_normal(h.shape[-1])).astype(np.float32)
                mean = h.mean(axis=-1, keepdims=True, dtype=np.float32)
                var = np.mean((h - mean) ** 2, axis=-1, keepdims=True, dtype=np.float32)
                return ((h - mean) / np.sqrt(var + np.float32(self.eps)) * self.gamma + self.beta).astype(np.float32)

        class Dropout(Layer):
            def __init__(self, rate, **kw):
                pass

            def apply(self, h):                       # inference: identity
                return h

        class Flatten(Layer):
            def apply(self, h):
                return h.reshape(h.shape[0], -1)

        class Model:
            def __init__(self, inputs, outputs, name=None):
                self.outputs, self.name = outputs, name

            def __call__(self, x):
                return self.outputs.fn(x)

            def summary(self):
                return ""

        layers = types.SimpleNamespace(Conv1D=Conv1D, Dense=Dense, Dropout=Dropout, Flatten=Flatten, MultiHeadAttention=MultiHeadAttention,
                                       LayerNormalization=LayerNormalization)
        self.tf = types.SimpleNamespace(keras=types.SimpleNamespace(layers=layers),
                                        expand_dims=lambda x, axis: Sym(lambda v: np.expand_dims(x.fn(v), axis)),
                                        squeeze=lambda x, axis: Sym(lambda v: np.squeeze(x.fn(v), axis)))
        self.glb = {"tf": self.tf, "Input": Input, "Model": Model, "regularizers": types.SimpleNamespace(l2=lambda v: None), "print": lambda *a, **k: None}


def build_reference_conv1d_model(p_in, p_out, seed):
    """``utils.define_model_arch('conv1D')`` (utils.py:435-461) and ``NNs.conv1D_PCA`` (NNs.py:75-124) executed as written."""
    UTL = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py"
    NNS = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/NNs.py"
    arch = _method(_tree(UTL), None, "define_model_arch", {}, UTL)
    n_layers, width = arch("conv1D")
    stub = _KerasStub(seed)
    fn = _method(_tree(NNS), None, "conv1D_PCA", stub.glb, NNS)
    model = fn(n_layers, width, p_in, p_out, 0.1, 1e-4)       # train.py:568 argument order (dropout, L2: inert at inference)
    return model, stub


def build_reference_attention_model(p_in, p_out, seed):
    """``utils.define_model_arch('MLP_attention')`` (utils.py:455-457) and ``NNs.densePCA_attention`` (NNs.py:40-72) executed as
    written.  Its ``Input((int(PC_input),))`` is a plain vector: the stub's Input must not add an axis."""
    UTL = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py"
    NNS = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/NNs.py"
    arch = _method(_tree(UTL), None, "define_model_arch", {}, UTL)
    n_layers, width = arch("MLP_attention")
    stub = _KerasStub(seed)
    fn = _method(_tree(NNS), None, "densePCA_attention", stub.glb, NNS)
    model = fn(n_layers, width, p_in, p_out, 0.1, 1e-4)       # train.py argument order (dropout, L2: inert at inference)
    return model, stub


def solid_band(grid, y0, y1, x0, x1):
    """Zero a rectangle of every channel (a big solid body): forces the
    'no flow cell in the strip' NaN branches of the reassembly."""
    g = grid.copy()
    g[y0:y1, x0:x1, :] = 0.0
    return g


# --------------------------------------------------------------------------
# the three reference executions
# --------------------------------------------------------------------------
def run_gradp(grid6, model, keep_labels=False):
    """UGP.timeStep, from the block extraction to the four assemblies."""
    tree = _tree(UGP)
    glb = {"np": np, "ndimage": None}
    asm = _method(tree, "Evaluation", "assemble_prediction", glb, UGP)
    Ev = type("Ev", (), {"assemble_prediction": asm})
    self = Ev()
    self.avance, self.shape = model.ov if model.ov is not None else int(0.75 * model.S), model.S
    self.pcainput, self.pcap = sk_pca(model.comp_in, model.mean_in), sk_pca(model.comp_out, model.mean_out)
    self.pc_in, self.pc_p = model.p_in, model.p_out
    self.max_abs_input_PCA, self.max_abs_output_PCA = model.in_a, model.out_a
    self.model = dense_callable(model.weights)
    body = _find_fn(tree, "timeStep", "Evaluation").body
    stmts = _slice(body, lambda s: s.startswith("x_list = []"), lambda s: s.startswith("test_dPdy ="))
    loc = {"self": self, "grid": grid6[None].astype(np.float64).copy(), "apply_filter": False}
    with np.errstate(all="ignore"):
        _run(stmts, glb, loc, UGP)
    out = dict(x_input=np.asarray(loc["x_input"], np.float64),
               fields=np.stack([loc["res_dPdx"][0, :, :, 0], loc["res_dPdy"][0, :, :, 0]], -1),
               n_blocks=np.int64(loc["N"]))
    if keep_labels:   # labels pushed through the same reassembly (UGP:546-547 self-check)
        out["label_fields"] = np.stack([loc["test_dPdx"][0, :, :, 0], loc["test_dPdy"][0, :, :, 0]], -1)
    return out


def run_deltas(grid5, model, U_max_norm=1.0, max_abs_p=1.0, _keep=None, keras_model=None, block_error=False):
    """SMD.timeStep, from the block extraction to ``assemble_prediction``."""
    tree = _tree(SMD)
    glb = {"np": np, "ndimage": None}
    asm = _method(tree, "Evaluation", "assemble_prediction", glb, SMD)
    Ev = type("Ev", (), {"assemble_prediction": asm})
    self = Ev()
    self.overlap, self.shape = model.ov if model.ov is not None else int(0.25 * model.S), model.S
    self.pcainput, self.pcap = sk_pca(model.comp_in, model.mean_in), sk_pca(model.comp_out, model.mean_out)
    self.pc_in, self.pc_p = model.p_in, model.p_out
    self.standardization_method = model.scaler_kind
    self.max_abs_input_PCA, self.max_abs_output_PCA = model.in_a, model.out_a
    self.max_abs_p = max_abs_p
    self.model = keras_model if keras_model is not None else dense_callable(model.weights)
    body = _find_fn(tree, "timeStep", "Evaluation").body
    stmts = _slice(body, lambda s: s.startswith("x_list = []"),
                   # block_error: stop BEFORE the assembly, which corrects the blocks of res_concat in place (a view is passed)
                   (lambda s: s.startswith("res_concat = res_concat * self.max_abs_p")) if block_error else
                   (lambda s: s.startswith("(deltap_res, change_in_deltap) =") or s.startswith("deltap_res, change_in_deltap =")))
    Ny, Nx = grid5.shape[:2]
    loc = {"self": self, "grid": grid5[None].astype(np.float64).copy(), "apply_filter": False,
           "U_max_norm": U_max_norm, "deltaU_change_grid": np.zeros((Ny, Nx)),
           "deltaP_prev_grid": np.zeros((Ny, Nx))}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            if model.scaler_kind == "std":
                np.savez("mean_std.npz", mean_in=model.in_a, std_in=model.in_b, mean_out=model.out_a, std_out=model.out_b)
            elif model.scaler_kind == "min_max":
                np.savez("min_max_values.npz", min_in=model.in_a, max_in=model.in_b, min_out=model.out_a, max_out=model.out_b)
            with np.errstate(all="ignore"):
                _run(stmts, glb, loc, SMD)
        finally:
            os.chdir(cwd)
            if _keep is not None:                        # what was computed before an exception of the reassembly
                _keep.update({k: loc[k] for k in ("x_input", "N") if k in loc})
    if block_error:
        # utils.compute_in_block_error (utils.py:210-243) itself, called with the arguments of SM_call.py:554-555 (the statement
        # is dropped from the slice above because it lives in another module): decoded, dimensional blocks against the de-meaned
        # label blocks times the same scale, over the blocks' flow cells
        UTL = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py"
        fn = _method(_tree(UTL), None, "compute_in_block_error", {"np": np, "print": lambda *a, **k: None}, UTL)
        flow_bool = self.x_array[..., 2:3] != 0
        a, b = fn(loc["res_concat"], self.y_array * self.max_abs_p * pow(U_max_norm, 2.0), flow_bool)
        return np.array([a, b], np.float64)
    return dict(x_input=np.asarray(loc["x_input"], np.float64), fields=np.asarray(loc["deltap_res"])[..., None],
                n_blocks=np.int64(loc["N"]))


def run_chapter5(grid3, model):
    """PM.py_func (rank-0 branch), from the block extraction to the final shift."""
    tree = _tree(PM)
    fn = _find_fn(tree, "py_func")
    rank0 = [n for n in fn.body if isinstance(n, ast.If) and "rank == 0" in _src(n.test)]
    body = max(rank0, key=lambda n: len(n.body)).body
    stmts = _slice(body, lambda s: s.startswith("x_list = []"), lambda s: s.startswith("result_array = result_array[0, :, :, 0]"))
    import time as _time
    glb = {"np": np, "time": _time, "model": dense_callable(model.weights),
           "pca_mean_input": model.mean_in, "comp_input": model.comp_in, "max_abs_input_PCA": model.in_a,
           "max_abs_p_PCA": model.out_a, "comp_p": model.comp_out, "pca_mean_p": model.mean_out}
    loc = {"grid": grid3[None].astype(np.float64).copy()}
    with np.errstate(all="ignore"):
        _run(stmts, glb, loc, PM)
    return dict(x_input=np.asarray(loc["x_input"], np.float64), fields=np.asarray(loc["result_array"])[..., None],
                n_blocks=np.int64(loc["N"]))


def run_domain_dist(top, obst, xy0):
    """python_module.py:72-99 `domain_dist`, executed: matplotlib's Path.contains_points and SciPy's cdist are the
    real ones; shapely (not installed) is replaced by a DATA stand-in that hands over the convex hull's exterior ring
    (SciPy qhull vertices in GEOS's clockwise closed order)."""
    import matplotlib.path as mpltPath
    from scipy.spatial import distance
    from psm_amd import geometry

    class MultiPoint:
        def __init__(self, pts):
            ring = geometry.convex_hull_ring(np.asarray(pts))
            self.convex_hull = types.SimpleNamespace(exterior=types.SimpleNamespace(coords=types.SimpleNamespace(xy=(ring[:, 0], ring[:, 1]))))
    glb = {"np": np, "MultiPoint": MultiPoint, "mpltPath": mpltPath, "distance": distance}
    fn = _find_fn(_tree(PM), "domain_dist")
    mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
    exec(compile(mod, PM, "exec"), glb)
    dom, sdf = glb["domain_dist"](top, obst, xy0)
    return np.asarray(dom, bool), np.asarray(sdf, np.float64)


def run_interp_weights_idw(xyz, uvw):
    """pressureSM_deltas/utils.py:22-55 `interp_weights` with its IDW fallback, executed (the file reaches KDTree
    through the name `sklearn`, which it never imports: injected here)."""
    import sklearn
    import sklearn.neighbors  # noqa: F401
    import scipy.spatial as _sp
    try:
        import scipy.spatial.qhull as qhull
    except Exception:
        qhull = types.SimpleNamespace(Delaunay=_sp.Delaunay)
    UTL = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py"
    fn = _find_fn(_tree(UTL), "interp_weights")
    mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
    glb = {"np": np, "qhull": qhull, "sklearn": sklearn}
    exec(compile(mod, UTL, "exec"), glb)
    v, w = glb["interp_weights"](xyz, uvw)
    return np.asarray(v, np.int64), np.asarray(w, np.float64)


C4E = f"{REF}/Thesis_Work/Chapter4/MLP/M_u/Evaluation/Eval_dual_Dense_onlycil.py"
C4F = f"{REF}/Thesis_Work/Chapter4/MLP/M_fU/Evaluation/Eval.py"
C4_MODELS = {"M_u": f"{REF}/Thesis_Work/Chapter4/MLP/M_u/trained_models/cil.h5",
             "M_fU": f"{REF}/Thesis_Work/Chapter4/MLP/M_fU/Evaluation/model_first_.h5"}


def run_chapter4(grid, model, which):
    """Thesis_Work/Chapter4 evaluators (M_u/Evaluation/Eval_dual_Dense_onlycil.py:236-414, M_fU/Evaluation/Eval.py:
    227-411): `timeStep` from the block extraction to the final shift, avance = int(0.75 * shape)."""
    path = C4E if which == "M_u" else C4F
    tree = _tree(path)
    body = _find_fn(tree, "timeStep", "Evaluation").body
    stmts = _slice(body, lambda s: s.startswith("x_list = []"), lambda s: s.startswith("result_array -= np.mean("))
    me = types.SimpleNamespace(avance=model.ov, shape=model.S, pcainput=sk_pca(model.comp_in, model.mean_in),
                               pcap=sk_pca(model.comp_out, model.mean_out), pc_in=model.p_in, pc_p=model.p_out,
                               max_abs_input_PCA=model.in_a, max_abs_p_PCA=model.out_a, model=dense_callable(model.weights))
    loc = {"self": me, "grid": grid[None].astype(np.float64).copy()}
    with np.errstate(all="ignore"):
        _run(stmts, {"np": np}, loc, path)
    return dict(x_input=np.asarray(loc["x_input"], np.float64), fields=np.asarray(loc["result_array"])[0],
                n_blocks=np.int64(loc["N"]))


def run_py_func_mesh(array, geo, model, maxs):
    """PM.py_func, the whole rank-0 body from the gathered cell array to the final p
    (python_module.py:264-496), with the one-time tables of init_func supplied (their
    construction needs shapely, which is not installed)."""
    tree = _tree(PM)
    import time as _time
    glb = {"np": np, "time": _time}
    for fname in ("interpolate", "interpolate_fill"):            # the reference's own helpers
        fn = _find_fn(tree, fname)
        mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
        exec(compile(mod, PM, "exec"), glb)
    glb.update(model=dense_callable(model.weights), pca_mean_input=model.mean_in, comp_input=model.comp_in,
               max_abs_input_PCA=model.in_a, max_abs_p_PCA=model.out_a, comp_p=model.comp_out, pca_mean_p=model.mean_out,
               max_abs_Ux=maxs[0], max_abs_Uy=maxs[1], max_abs_dist=maxs[2], max_abs_p=maxs[3],
               vert_OFtoNP=geo.vert_m2g, weights_OFtoNP=geo.wts_m2g, vert_NPtoOF=geo.vert_g2m, weights_NPtoOF=geo.wts_g2m,
               indices=geo.indices, sdfunct=geo.sdfunct[:, :, None], grid_shape_y=geo.ny, grid_shape_x=geo.nx)
    fn = _find_fn(tree, "py_func")
    rank0 = [n for n in fn.body if isinstance(n, ast.If) and "rank == 0" in _src(n.test)]
    body = max(rank0, key=lambda n: len(n.body)).body
    stmts = _slice(body, lambda s: s.startswith("array = np.concatenate(array_global)"),
                   lambda s: s.startswith("p[np.isnan(p_interp)] ="))
    loc = {"array_global": [np.asarray(array, np.float64)]}
    with np.errstate(all="ignore"):
        _run(stmts, glb, loc, PM)
    return dict(p=np.asarray(loc["p"], np.float64), grid=np.asarray(loc["grid"][0], np.float64),
                U_max_norm=np.float64(loc["U_max_norm"]), n_blocks=np.int64(loc["N"]))


def run_init_helpers(array, delta):
    """The pure NumPy/SciPy helpers of init_func executed from the reference file:
    create_uniform_grid (PM:42-48) and interp_weights (PM:154-161) in both directions."""
    tree = _tree(PM)
    import scipy.spatial as _sp
    try:
        import scipy.spatial.qhull as qhull          # the reference's import (older SciPy)
    except Exception:
        qhull = types.SimpleNamespace(Delaunay=_sp.Delaunay)
    glb = {"np": np, "qhull": qhull, "d": 2}
    for fname in ("create_uniform_grid",):
        fn = _find_fn(tree, fname)
        mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
        exec(compile(mod, PM, "exec"), glb)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "interp_weights"]
    mod = ast.Module(body=[fns[-1]], type_ignores=[]); ast.fix_missing_locations(mod)   # the later definition wins at import
    exec(compile(mod, PM, "exec"), glb)
    x_min, x_max = round(np.min(array[:, 2]), 2), round(np.max(array[:, 2]), 2)
    y_min, y_max = round(np.min(array[:, 3]), 2), round(np.max(array[:, 3]), 2)
    X0, Y0 = glb["create_uniform_grid"](x_min, x_max, y_min, y_max, delta)
    xy0 = np.concatenate((np.expand_dims(X0, axis=1), np.expand_dims(Y0, axis=1)), axis=-1)
    v1, w1 = glb["interp_weights"](array[:, 2:4], xy0)
    v2, w2 = glb["interp_weights"](xy0, array[:, 2:4])
    return X0, Y0, v1, w1, v2, w2


# --------------------------------------------------------------------------
# cases (inputs are regenerated from these specs by tests/cases.py)
# --------------------------------------------------------------------------
def run_poisson_features(c):
    """pressureSM_Poisson/SM_call.py timeStep: the nested gradient function, the NaN masking, the Poisson
    source term, `smart_arcsin_smooth_transform` and the grid fill / rescale (SMP:602-646, 696-710)."""
    SMP = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_Poisson/SM_call.py"
    tree = _tree(SMP)
    glb = {"np": np}
    fn = _find_fn(tree, "smart_arcsin_smooth_transform")
    mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
    exec(compile(mod, SMP, "exec"), glb)
    body = _find_fn(tree, "timeStep", "Evaluation").body
    drop = DROP + ("axes[", "im1 =", "im2 =", "import matplotlib", "(fig, axes)", "fig, axes")
    def sl(first, last):
        i0 = next(i for i, st in enumerate(body) if _src(st).startswith(first))
        i1 = next(i for i, st in enumerate(body) if i >= i0 and _src(st).startswith(last))
        return [st for st in body[i0:i1 + 1] if not any(d in _src(st) for d in drop)]
    Ny, Nx = c["ux"].shape
    me = types.SimpleNamespace(sdfunct=c["sdfunct"][:, :, None].copy(), k=c["k"], grid_shape_y=Ny, grid_shape_x=Nx,
                               max_abs_Poisson_term_1=c["max_abs"][0], max_abs_delta_Ux=c["max_abs"][1],
                               max_abs_delta_Uy=c["max_abs"][2], max_abs_dist=c["max_abs"][3], max_abs_delta_p=1.0)
    loc = {"self": me, "ux_grid": c["ux"].copy(), "uy_grid": c["uy"].copy(), "delta_ux_grid": c["dux"].copy(),
           "delta_uy_grid": c["duy"].copy(), "L": c["L"], "U": c["U"],
           "delta_p_grid": np.zeros((Ny, Nx)), "p_grid": np.zeros((Ny, Nx))}
    with np.errstate(all="ignore"):
        _run(sl("def gradient_with_nan_direct_neighbors", "Poisson_term_1_f = smart_arcsin_smooth_transform"), glb, loc, SMP)
        _run(sl("grid = np.zeros(shape=(1, self.grid_shape_y, self.grid_shape_x, 6))", "grid[0, :, :, 4] /= self.max_abs_delta_p"), glb, loc, SMP)
    term = np.asarray(loc["Poisson_term_1"], np.float64)
    return dict(grid=np.asarray(loc["grid"][0, :, :, :4], np.float32), term_crop=term[48:112, 40:104].copy(),
                term_sum=np.float64(term.sum()), term_abs_sum=np.float64(np.abs(term).sum()))


def run_evaluator_grid(cells, tables, maxs):
    """pressureSM_deltas/SM_call.py timeStep from the column split to the weighting images (:381-451), with the
    reference's own `utils.interpolate_fill` (utils.py:75-90); the interpolation tables come from the build's
    geometry step (computeOnlyOnce needs shapely)."""
    UTL = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py"
    ns = {"np": np}
    fn = _find_fn(_tree(UTL), "interpolate_fill")
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(mod)
    exec(compile(mod, UTL, "exec"), ns)
    utils = types.SimpleNamespace(interpolate_fill=ns["interpolate_fill"])
    body = _find_fn(_tree(SMD), "timeStep", "Evaluation").body
    i0 = next(i for i, st in enumerate(body) if _src(st).startswith("i = 0"))
    i1 = next(i for i, st in enumerate(body) if _src(st).startswith("deltaP_prev_grid[tuple(self.indices.T)]"))
    stmts = [st for st in body[i0:i1 + 1] if "print(" not in _src(st)]
    me = types.SimpleNamespace(indice=cells.shape[0], vert=tables.vtx_m2g, weights=tables.wts_m2g, indices=tables.indices,
                               sdfunct=tables.sdfunct[:, :, None], grid_shape_y=tables.ny, grid_shape_x=tables.nx,
                               max_abs_Ux=maxs[0], max_abs_Uy=maxs[1], max_abs_dist=maxs[2], max_abs_p=maxs[3])
    loc = {"self": me, "data": cells[None, None].astype(np.float32)}
    with np.errstate(all="ignore"):
        _run(stmts, {"np": np, "utils": utils, "pow": pow}, loc, SMD)
    g = np.asarray(loc["grid"][0], np.float64)
    return dict(grid_crop=g[40:100, 120:200].copy(), grid_sum=g.sum(axis=(0, 1)), grid_abs_sum=np.abs(g).sum(axis=(0, 1)),
                dU_crop=np.asarray(loc["deltaU_change_grid"])[40:100, 120:200].copy(),
                dPprev_crop=np.asarray(loc["deltaP_prev_grid"])[40:100, 120:200].copy(), U_max_norm=np.float64(loc["U_max_norm"]))


def run_evaluator_grid_gradp(cells32, tables, maxs, ext):
    """Eval_dual_Dense_onlycil.py timeStep from the column split to the rescaled 6-channel grid (:429-467), with the
    class's own `interpolate_fill` method; tables from the build's geometry step."""
    tree = _tree(UGP)
    interp = _method(tree, "Evaluation", "interpolate_fill", {"np": np}, UGP)
    Ev = type("Ev", (), {"interpolate_fill": interp})
    me = Ev()
    me.indice, me.vert, me.weights, me.indices = cells32.shape[0], tables.vtx_m2g, tables.wts_m2g, tables.indices
    me.sdfunct, me.grid_shape_y, me.grid_shape_x = tables.sdfunct[:, :, None], tables.ny, tables.nx
    me.min_x, me.max_x, me.min_y, me.max_y = ext
    me.max_abs_Ux, me.max_abs_Uy, me.max_abs_dist, me.max_abs_dPdx, me.max_abs_dPdy = maxs
    body = _find_fn(tree, "timeStep", "Evaluation").body
    i0 = next(i for i, st in enumerate(body) if _src(st).startswith("i = 0"))
    i1 = next(i for i, st in enumerate(body) if _src(st).startswith("grid[0, :, :, 4:5] = grid[0, :, :, 4:5] / self.max_abs_dPdy"))
    stmts = [st for st in body[i0:i1 + 1] if "print(" not in _src(st)]
    loc = {"self": me, "data": cells32[None, None]}
    with np.errstate(all="ignore"):
        _run(stmts, {"np": np, "pow": pow}, loc, UGP)
    g = np.asarray(loc["grid"][0], np.float64)
    return dict(grid_crop=g[170:230, 60:140].copy(), grid_sum=g.sum(axis=(0, 1)), grid_abs_sum=np.abs(g).sum(axis=(0, 1)),
                U_max_norm=np.float64(loc["U_max_norm"]))


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    os.makedirs(HERE, exist_ok=True)

    # real trained artefacts of the reference test case -> data fixture
    W = formats.read_keras_dense_weights(WEIGHTS)
    maxs, maxs_pca = formats.read_maxs(MAXS), formats.read_maxs(MAXS_PCA)
    flat = {}
    for i, (w, b) in enumerate(W):
        flat[f"W{i}"], flat[f"b{i}"] = w, b
    np.savez_compressed(os.path.join(HERE, "chapter5_weights.npz"), maxs=maxs, maxs_PCA=maxs_pca, **flat)
    print("chapter5_weights.npz", [w.shape for w, _ in W])

    # trained Chapter-4 networks (read with the build's own HDF5 reader) + the M_fU evaluator's scale files
    flat = {}
    for which, path in C4_MODELS.items():
        for i, (w, b) in enumerate(formats.read_keras_dense_weights(path)):
            flat[f"{which}_W{i}"], flat[f"{which}_b{i}"] = w, b
    c4dir = f"{REF}/Thesis_Work/Chapter4/MLP/M_fU/Evaluation"
    np.savez_compressed(os.path.join(HERE, "chapter4_weights.npz"), MfU_maxs=formats.read_maxs(f"{c4dir}/maxs"),
                        MfU_maxs_PCA=formats.read_maxs(f"{c4dir}/maxs_PCA"), **flat)
    print("chapter4_weights.npz", sorted(k for k in flat if k.endswith("W0")), [flat[k].shape for k in sorted(flat) if "_W" in k])

    # ---- domain_dist (a3) and the IDW fallback of the Improved_SM interp_weights (a2)
    dd = {}
    for shape in ("circle", "rectangle"):
        top, obst, xy0 = cases.build_domain_case(shape)
        dom, sdf = run_domain_dist(top, obst, xy0)
        dd[f"{shape}_domain"] = np.packbits(dom)
        dd[f"{shape}_n"] = np.int64(len(dom))
        dd[f"{shape}_sdf_tail"] = sdf[-400:].copy()                      # the hand-placed hard points
        dd[f"{shape}_sdf_sum"] = np.float64(sdf.sum())
        dd[f"{shape}_sdf_crop"] = sdf[:42000].reshape(140, 300)[40:100, 40:120].copy()
        print(f"domain_dist {shape}: inside domain {int(dom.sum())} of {len(dom)}")
    np.savez_compressed(os.path.join(HERE, "domain_dist_case.npz"), **dd)
    xyz, uvw = cases.build_idw_case()
    v, w = run_interp_weights_idw(xyz, uvw)
    np.savez_compressed(os.path.join(HERE, "interp_weights_idw.npz"), vertices=v.astype(np.int32), weights=w)
    print("interp_weights_idw: targets", len(uvw), "outside hull (all weights >= 0 after the fallback):", int((w >= 0).all(axis=1).sum()))

    # ---- mesh-side boundary (py_func on one rank) --------------------------------
    from oracle import psm_oracle as orc
    array, top, obst, model, maxs = cases.build_mesh_case()
    geo = orc.init_geometry(array, top, obst)
    X0, Y0, v1, w1, v2, w2 = run_init_helpers(array, 5e-3)
    out = run_py_func_mesh(array, geo, model, maxs)
    # table checks kept small: checksums of the reference helpers' outputs
    out.update(ref_grid_n=np.int64(len(X0)), ref_X0_sum=np.float64(X0.sum()), ref_Y0_sum=np.float64(Y0.sum()),
               ref_v1_sum=np.int64(v1.astype(np.int64).sum()), ref_v2_sum=np.int64(v2.astype(np.int64).sum()),
               ref_w1_abs_sum=np.float64(np.abs(w1).sum()), ref_w2_abs_sum=np.float64(np.abs(w2).sum()))
    out["grid"] = out["grid"].astype(np.float32)          # keep the fixture small; p stays float64
    np.savez_compressed(os.path.join(HERE, "mesh_chapter5.npz"), **out)
    print(f"mesh_chapter5: N={len(array)} grid={geo.ny}x{geo.nx} B={int(out['n_blocks'])} |p|max={np.abs(out['p']).max():.4f} "
          f"fallback cells={(out['p'] == array[:, 4]).sum()}")

    # ---- optional post-steps of assemble_prediction (pressureSM_Poisson/SM_call.py:334, 504-517):
    #      Gaussian filter + deltaU-change weighting, run on the decoded blocks of one deltas case
    grid, model, bp, dU, dPprev = cases.build_filter_case()
    SMP = f"{REF}/Improved_SM/deltaU_to_deltaP/source/pressureSM_Poisson/SM_call.py"
    import scipy.ndimage as _ndi
    tree = _tree(SMP)
    asm = _method(tree, "Evaluation", "assemble_prediction", {"np": np, "ndimage": _ndi}, SMP)
    lay = orc.block_layout("deltas", grid.shape[0], grid.shape[1])
    me = types.SimpleNamespace(overlap=lay.ov, shape=lay.S, Ref_BC=0, x_array=orc.extract_blocks(grid, lay, 3))
    with np.errstate(all="ignore"):
        res, chg = asm(me, bp.copy(), [list(t) for t in lay.tags], lay.n_x, lay.n_y, True, grid.shape[1], grid.shape[0],
                       dU.copy(), dPprev.copy(), True)
    np.savez_compressed(os.path.join(HERE, "deltas_filters_256x256.npz"), result=res.astype(np.float32), change=chg.astype(np.float32))
    print("deltas_filters_256x256: |res|max=%.4f |change|max=%.5f" % (np.abs(res).max(), np.abs(chg).max()))

    # ---- U_to_gradP: integrate_field + four-quadrant stitching (UGP:371-416, 592-628)
    ic = cases.build_integration_case()
    tree = _tree(UGP)
    integ = _method(tree, "Evaluation", "integrate_field", {"np": np}, UGP)
    Ev = type("Ev", (), {"integrate_field": integ})
    me = Ev()
    me.sdfunct = ic["sdfunct"][:, :, None]; me.X0 = ic["X0"]; me.delta = ic["delta"]
    me.min_x, me.max_x, me.min_y, me.max_y = ic["min_x"], ic["max_x"], ic["min_y"], ic["max_y"]
    body = _find_fn(tree, "timeStep", "Evaluation").body
    stmts = _slice(body, lambda s: s.startswith("gradP = np.concatenate"), lambda s: s.startswith("result[center_p_y:, :center_p_x] = pBlock4Corrected"))
    g4 = ic["gradP"][None]
    loc = {"self": me, "res_dPdx": g4[..., 0:1].copy(), "res_dPdy": g4[..., 1:2].copy(), "grid": np.zeros((1,) + ic["gradP"].shape[:2] + (6,))}
    with np.errstate(all="ignore"):
        _run(stmts, {"np": np}, loc, UGP)
    np.savez_compressed(os.path.join(HERE, "gradp_integration_320x384.npz"), p=np.asarray(loc["result"], np.float32),
                        center_p_x=np.int64(loc["center_p_x"]), center_p_y=np.int64(loc["center_p_y"]))
    print("gradp_integration: center", int(loc["center_p_x"]), int(loc["center_p_y"]), "|p|max=%.4f" % np.abs(loc["result"]).max())

    # ---- dataset-driven evaluator front end (SMD:381-451) on frame 1 of the synthetic dataset
    from psm_amd import geometry
    with tempfile.TemporaryDirectory() as td:
        dc = cases.build_dataset_case(td)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    cells0 = np.asarray(dc["sim"][0, 0, :dc["N"]], np.float64)
    tabs = geometry.build_geometry_evaluator(cells0[:, 3:5], cells0[:, 2], f32(dc["top"]), f32(dc["obst"]), 5e-3, idw_fallback=True)
    out = run_evaluator_grid(np.asarray(dc["sim"][0, 1, :dc["N"]], np.float64), tabs, cases.DATASET_MAXS)
    np.savez_compressed(os.path.join(HERE, "evaluator_grid_138x300.npz"), **out)
    print("evaluator_grid: U_max_norm=%.6f |grid| sums" % out["U_max_norm"], out["grid_abs_sum"])

    # ---- U_to_gradP evaluator front end (UGP:429-467) on frame 1 of its synthetic dataset
    with tempfile.TemporaryDirectory() as td:
        gc = cases.build_gradp_dataset_case(td)
    cells0 = np.asarray(gc["sim"][0, 0, :gc["N"]], np.float64)
    top32 = np.asarray(gc["top"], np.float32)
    tabs = geometry.build_geometry_evaluator(cells0[:, 3:5], cells0[:, 5], f32(gc["top"]), f32(gc["obst"]), 5e-3, every=2,
                                             round_digits=2, box="top")
    ext = (np.min(top32[:, 0]), np.max(top32[:, 0]), np.min(top32[:, 1]), np.max(top32[:, 1]))
    out = run_evaluator_grid_gradp(gc["sim"][0, 1, :gc["N"]], tabs, cases.GRADP_MAXS, ext)
    np.savez_compressed(os.path.join(HERE, "evaluator_grid_gradp_320x300.npz"), **out)
    print("evaluator_grid_gradp: U_max_norm=%.6f |grid| sums" % out["U_max_norm"], out["grid_abs_sum"])

    # ---- pressureSM_Poisson feature builder
    out = run_poisson_features(cases.build_poisson_case())
    np.savez_compressed(os.path.join(HERE, "poisson_features_160x200.npz"), **out)
    print("poisson_features: |grid|max per channel", np.abs(out["grid"]).max(axis=(0, 1)))

    # ---- p_i == 0 grids (BASELINE configs[1] / configs[4] shapes): what the reference's statements do there
    for name, sp in cases.DEGENERATE_CASES.items():
        grid, model = cases.build(name)
        if sp["variant"] == "gradp":
            out = run_gradp(grid, model)
            f = out["fields"]
            out["raised"] = np.int64(0)
            print(f"{name}: B={int(out['n_blocks'])} NaN cells {int(np.isnan(f).sum())} of {f.size}")
        else:
            keep = {}
            try:
                run_deltas(grid, model, _keep=keep)
                raised = 0
            except ValueError as e:                      # 'could not broadcast input array ...' (SM_call.py:335)
                raised = 1
                print(f"{name}: the reference raises {type(e).__name__}")
            out = dict(raised=np.int64(raised), x_input=np.asarray(keep["x_input"], np.float64), n_blocks=np.int64(keep["N"]))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)

    # ---- compute_in_block_error (utils.py:210-243 at SM_call.py:555): the block-level error of three deltas cases, labels = the
    #      grids' synthetic delta-p channel
    be = {}
    for name in cases.BLOCK_ERROR_CASES:
        grid, model = cases.build(name)
        be[name] = run_deltas(grid, model, U_max_norm=cases.GOLDEN_CASES[name].get("U_max_norm", 1.0),
                              max_abs_p=cases.GOLDEN_CASES[name].get("max_abs_p", 1.0), block_error=True)
        print(f"block_error {name}: pred_minus_true_block={be[name][0]:.6e} squared={be[name][1]:.6e}")
    np.savez_compressed(os.path.join(HERE, "block_error_deltas.npz"), **be)

    for name in cases.GOLDEN_CASES:
        grid, model = cases.build(name)
        if cases.GOLDEN_CASES[name].get("chapter4"):
            out = run_chapter4(grid, model, cases.GOLDEN_CASES[name]["chapter4"])
        elif model.variant == "gradp":
            out = run_gradp(grid, model, keep_labels=(name == "gradp_272x288"))
        elif cases.GOLDEN_CASES[name].get("conv1d"):
            km, stub = build_reference_conv1d_model(model.p_in, model.p_out, seed=cases.GOLDEN_CASES[name]["seed"])
            out = run_deltas(grid, model, keras_model=km)
            convs = [l for l in stub.layers if hasattr(l, "K")]
            dense = [l for l in stub.layers if hasattr(l, "W")]
            assert len(dense) == 1 and stub.layers.index(dense[0]) == len(stub.layers) - 1
            for i, l in enumerate(convs):
                out[f"convK{i}"], out[f"convb{i}"] = l.K, l.b
            out["denseW"], out["denseb"] = dense[0].W, dense[0].b
            print(f"{name}: reference-built conv1D_PCA: filters {[l.filters for l in convs]}, kernel {convs[0].k}, head {dense[0].W.shape}")
        elif cases.GOLDEN_CASES[name].get("attention"):
            km, stub = build_reference_attention_model(model.p_in, model.p_out, seed=cases.GOLDEN_CASES[name]["seed"])
            out = run_deltas(grid, model, keras_model=km)
            kinds = [type(l).__name__ for l in stub.layers]
            dense = [l for l in stub.layers if hasattr(l, "W")]
            mha = [l for l in stub.layers if kinds[stub.layers.index(l)] == "MultiHeadAttention"]
            lns = [l for l in stub.layers if hasattr(l, "gamma")]
            assert len(mha) == 1 and len(lns) == len(dense) - 1 and kinds[0] == "Dense" and kinds[-1] == "Dense", kinds
            for i, l in enumerate(dense):
                out[f"denseW{i}"], out[f"denseb{i}"] = l.W, l.b
            for k, v in mha[0].P.items():
                if k[1] in "vo":      # the query / key projections cannot change the result (softmax over ONE key, tests/test_attention.py): not kept
                    out["att_" + k] = v
            for i, l in enumerate(lns):
                out[f"ln_gamma{i}"], out[f"ln_beta{i}"] = l.gamma, l.beta
            out["ln_eps"] = np.float64(lns[0].eps)
            print(f"{name}: reference-built densePCA_attention: layers {kinds}, dense {[l.W.shape for l in dense]}, heads {mha[0].h} x {mha[0].kd}")
        elif model.variant == "deltas":
            out = run_deltas(grid, model, U_max_norm=cases.GOLDEN_CASES[name].get("U_max_norm", 1.0),
                             max_abs_p=cases.GOLDEN_CASES[name].get("max_abs_p", 1.0))
        else:
            out = run_chapter5(grid, model)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        f = out["fields"]
        print(f"{name}: B={int(out['n_blocks'])} fields{f.shape} nan={int(np.isnan(f).sum())} "
              f"|x_input|max={np.abs(out['x_input']).max():.3f} |f|max={np.nanmax(np.abs(f)):.4f}")


if __name__ == "__main__":
    main()
