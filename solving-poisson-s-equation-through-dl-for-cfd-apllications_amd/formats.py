"""On-disk artefact readers for the surrogate path.

* Keras ``.h5`` weight files (``weights.h5``, ``model_*.h5``): HDF5 superblock
  v0, old-style groups (B-tree + local heap + symbol nodes), v1 object
  headers, contiguous little-endian datasets.  Written by the reference with
  ``model.save_weights`` (Thesis_Work/Chapter5/parallelized/test_case/
  save_weights.py:1-4) and read with ``model.load_weights`` (python_module.py:170).
  h5py is not available on the target image, so this is a small dependency-free
  reader for exactly that subset; anything else raises ``H5FormatError``.
* ``maxs`` / ``maxs_PCA`` text files (python_module.py:106-110, SM_call.py:70-72).
* ``mean_std.npz`` / ``min_max_values.npz`` scalers (utils.py:299,313).
* The padded simulation dataset (``sim_data[sim, t, max_cells, C]``, ``top_bound`` / ``obst_bound``
  ``[sim, t, max_points, 2]``, float32, contiguous, pad value -100.0; written by
  data_generation.py:64-74 with ``create_dataset(name, shape, np.float32)``, read by
  utils.read_dataset, utils.py:57-71): memory-mapped, one frame copied per request.
* PCA artefacts: the reference pickles scikit-learn / dask-ml ``IncrementalPCA`` objects
  (``ipca_input.pkl``, ``ipca_p.pkl``; SM_call.py:81-82); ``load_pca`` reads such a pickle (needs
  scikit-learn importable, as the reference does) or the dependency-free ``.npz`` export
  (``components_``, ``mean_``, ``explained_variance_ratio_``) written by ``save_pca_npz``.
"""
from __future__ import annotations

import mmap
import os
import pickle
import re
import struct
from typing import Dict, List, Tuple

import numpy as np


class H5FormatError(ValueError):
    pass


_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _H5:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != _SIG:
            raise H5FormatError("not an HDF5 file")
        ver = buf[8]
        if ver != 0:
            raise H5FormatError(f"superblock version {ver} not supported (only v0)")
        self.so, self.sl = buf[13], buf[14]
        if (self.so, self.sl) != (8, 8):
            raise H5FormatError("only 8-byte offsets/lengths supported")
        # 8 sig + 8 version bytes + 2+2 (group K) + 4 flags = 24 ; then 4 addresses
        self.base = self.u64(24)
        root_entry = 24 + 4 * 8
        self.root = self._sym_entry(root_entry)

    # -- primitives
    def u8(self, o): return self.b[o]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.b, o)[0]

    def _sym_entry(self, o):
        name_off = self.u64(o)
        hdr = self.u64(o + 8)
        cache = self.u32(o + 16)
        btree = heap = None
        if cache == 1:
            btree = self.u64(o + 24)
            heap = self.u64(o + 32)
        return dict(name_off=name_off, hdr=hdr, cache=cache, btree=btree, heap=heap)

    # -- object header v1: list of (type, body_offset, size)
    def messages(self, addr) -> List[Tuple[int, int, int]]:
        if self.u8(addr) != 1:
            raise H5FormatError("only version-1 object headers supported")
        nmsg = self.u16(addr + 2)
        hsize = self.u32(addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            o, size = blocks.pop(0)
            end = o + size
            while o + 8 <= end and len(out) < nmsg:
                mtype = self.u16(o)
                msize = self.u16(o + 2)
                body = o + 8
                if mtype == 0x10:                      # continuation
                    blocks.append((self.u64(body), self.u64(body + 8)))
                out.append((mtype, body, msize))
                o = body + msize
        return out

    def _heap_data(self, heap_addr) -> int:
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5FormatError("bad local heap")
        return self.u64(heap_addr + 8 + 8 + 8)

    def _cstr(self, o) -> str:
        e = self.b.find(b"\0", o)
        return bytes(self.b[o:e]).decode("ascii")

    def _group_tables(self, hdr_addr):
        """B-tree / heap addresses of an old-style group from its symbol-table message."""
        for mtype, body, _ in self.messages(hdr_addr):
            if mtype == 0x11:
                return self.u64(body), self.u64(body + 8)
        return None

    def children(self, btree, heap) -> Dict[str, dict]:
        names: Dict[str, dict] = {}
        hdata = self._heap_data(heap)

        def walk(node):
            if self.b[node:node + 4] == b"TREE":
                ntype, level, used = self.u8(node + 4), self.u8(node + 5), self.u16(node + 6)
                if ntype != 0:
                    raise H5FormatError("unexpected B-tree node type")
                o = node + 8 + 16          # skip siblings
                for k in range(used):
                    child = self.u64(o + 8 + k * 16)   # key(8) child(8) pairs
                    walk(child)
            elif self.b[node:node + 4] == b"SNOD":
                n = self.u16(node + 6)
                for k in range(n):
                    e = self._sym_entry(node + 8 + k * 40)
                    names[self._cstr(hdata + e["name_off"])] = e
            else:
                raise H5FormatError("bad group node")
        walk(btree)
        return names

    def dataset(self, hdr_addr, copy: bool = True) -> np.ndarray:
        shape = dtype = None
        data_addr = data_size = None
        for mtype, body, _ in self.messages(hdr_addr):
            if mtype == 0x01:                          # dataspace
                ver, rank = self.u8(body), self.u8(body + 1)
                o = body + (8 if ver == 1 else 4)
                shape = tuple(self.u64(o + 8 * i) for i in range(rank))
            elif mtype == 0x03:                        # datatype
                cls = self.u8(body) & 0x0F
                bits0 = self.u8(body + 1)
                size = self.u32(body + 4)
                if bits0 & 1:
                    raise H5FormatError("big-endian data not supported")
                if cls == 1 and size in (4, 8):
                    dtype = np.dtype("<f4" if size == 4 else "<f8")
                elif cls == 0 and size in (4, 8):
                    dtype = np.dtype("<i4" if size == 4 else "<i8")
                else:
                    dtype = None
            elif mtype == 0x08:                        # layout
                ver = self.u8(body)
                if ver == 3:
                    if self.u8(body + 1) != 1:
                        raise H5FormatError("only contiguous layout supported")
                    data_addr, data_size = self.u64(body + 2), self.u64(body + 10)
                elif ver in (1, 2):
                    rank, lclass = self.u8(body + 1), self.u8(body + 2)
                    if lclass != 1:
                        raise H5FormatError("only contiguous layout supported")
                    data_addr = self.u64(body + 8)
                    data_size = None
                else:
                    raise H5FormatError("layout version not supported")
        if shape is None or dtype is None or data_addr is None or data_addr == _UNDEF:
            raise H5FormatError("not a plain numeric contiguous dataset")
        count = int(np.prod(shape)) if shape else 1
        if self.base + data_addr + count * dtype.itemsize > len(self.b):
            raise H5FormatError("dataset extends past the end of the file")
        a = np.frombuffer(self.b, dtype=dtype, count=count, offset=self.base + data_addr)
        return a.reshape(shape).copy() if copy else a.reshape(shape)

    def walk(self, entry=None, prefix=""):
        """Yield (path, symbol entry) for every leaf that is not a group."""
        entry = self.root if entry is None else entry
        tabs = (entry["btree"], entry["heap"]) if entry["cache"] == 1 else self._group_tables(entry["hdr"])
        if tabs is None:
            yield prefix, entry
            return
        for name, e in self.children(*tabs).items():
            yield from self.walk(e, f"{prefix}/{name}")


def read_h5_datasets(path: str) -> Dict[str, np.ndarray]:
    """All numeric contiguous datasets of a (v0, contiguous) HDF5 file by path."""
    with open(path, "rb") as f:
        h = _H5(f.read())
    out = {}
    for p, e in h.walk():
        try:
            out[p] = h.dataset(e["hdr"])
        except H5FormatError:
            continue
    return out


def _layer_key(name: str):
    m = re.fullmatch(r"dense(?:_(\d+))?", name)
    return (0, int(m.group(1) or 0)) if m else (1, name)


def read_keras_dense_weights(path: str) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Ordered [(kernel[in,out] f32, bias[out] f32)] of a Keras Dense stack.

    Layers are identified by their HDF5 group names (``dense``, ``dense_1`` ...)
    and ordered by that index, then checked for chaining shapes -- the order in
    the file is never assumed (SURVEY.md §7 "hard parts")."""
    ds = read_h5_datasets(path)
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for p, a in ds.items():
        parts = [q for q in p.split("/") if q]
        leaf = parts[-1]
        if leaf not in ("kernel:0", "bias:0"):
            continue
        lname = next((q for q in parts if re.fullmatch(r"dense(?:_\d+)?", q)), None)
        if lname is None:
            continue
        layers.setdefault(lname, {})[leaf] = a
    names = sorted(layers, key=_layer_key)
    out = []
    for n in names:
        if "kernel:0" not in layers[n] or "bias:0" not in layers[n]:
            raise H5FormatError(f"layer {n} lacks kernel or bias")
        W = np.ascontiguousarray(layers[n]["kernel:0"], np.float32)
        b = np.ascontiguousarray(layers[n]["bias:0"], np.float32)
        if W.ndim != 2 or b.shape != (W.shape[1],):
            raise H5FormatError(f"layer {n}: unexpected shapes {W.shape} {b.shape}")
        out.append((W, b))
    for (W0, _), (W1, _) in zip(out, out[1:]):
        if W0.shape[1] != W1.shape[0]:
            raise H5FormatError("dense layers do not chain")
    if not out:
        raise H5FormatError("no dense layers found")
    return out


def read_keras_conv1d_head(path: str):
    """A Keras HDF5 file of the reference's NN zoo (NNs.py) -> (conv1d layers, dense layers): the Conv1D layers
    ``conv1d``, ``conv1d_1``, ... of ``conv1D_PCA`` (kernel [k, c_in, c_out]) in creation order -- an empty list for the
    Dense stacks of ``densePCA`` -- and the Dense layers behind them.  The first Dense layer of a conv1D_PCA file takes the
    flattened [p_in, filters] activation."""
    ds = read_h5_datasets(path)
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for p, a in ds.items():
        parts = [q for q in p.split("/") if q]
        if parts[-1] not in ("kernel:0", "bias:0"):
            continue
        lname = next((q for q in parts if re.fullmatch(r"conv1d(?:_\d+)?", q)), None)
        if lname is not None:
            layers.setdefault(lname, {})[parts[-1]] = a
    convs = []
    for n in sorted(layers, key=lambda s: int(s.split("_")[1]) if "_" in s else 0):
        if "kernel:0" not in layers[n] or "bias:0" not in layers[n]:
            raise H5FormatError(f"layer {n} lacks kernel or bias")
        K = np.ascontiguousarray(layers[n]["kernel:0"], np.float32)
        b = np.ascontiguousarray(layers[n]["bias:0"], np.float32)
        if K.ndim != 3 or b.shape != (K.shape[2],):
            raise H5FormatError(f"layer {n}: unexpected shapes {K.shape} {b.shape}")
        if convs and convs[-1][0].shape[2] != K.shape[1]:
            raise H5FormatError("conv1d layers do not chain")
        convs.append((K, b))
    if convs and convs[0][0].shape[1] != 1:
        raise H5FormatError("the first Conv1D layer of conv1D_PCA has one input channel")
    dense = read_keras_dense_weights(path)
    if convs and dense[0][0].shape[0] % convs[-1][0].shape[2] != 0:
        raise H5FormatError("the Dense layer behind the Conv1D stack does not take a flattened [p_in, filters] input")
    return convs, dense


def read_keras_attention(path: str):
    """The attention part of a ``densePCA_attention`` file (NNs.py:40-72), or None for a file without one: the
    ``MultiHeadAttention`` layer's EinsumDense weights -- groups ``query`` / ``key`` / ``value`` (kernel [d, heads, dim], bias
    [heads, dim]) and ``attention_output`` (kernel [heads, dim, d], bias [d]) below a group ``multi_head_attention`` -- and the
    ``LayerNormalization`` layers ``layer_normalization``, ``layer_normalization_1``, ... (``gamma:0``, ``beta:0``) in creation
    order.  -> the dict ``SurrogateModel.attention`` takes ({"Wq", "bq", "Wk", "bk", "Wv", "bv", "Wo", "bo", "ln", "eps"});
    ``eps`` is Keras' default 1e-3 (weight files do not carry it)."""
    ds = read_h5_datasets(path)
    mha: Dict[str, Dict[str, np.ndarray]] = {}
    lns: Dict[str, Dict[str, np.ndarray]] = {}
    for p, a in ds.items():
        parts = [q for q in p.split("/") if q]
        leaf = parts[-1]
        if any(re.fullmatch(r"multi_head_attention(?:_\d+)?", q) for q in parts) and leaf in ("kernel:0", "bias:0"):
            sub = next((q for q in parts if q in ("query", "key", "value", "attention_output")), None)
            if sub is not None:
                mha.setdefault(sub, {})[leaf] = a
        ln = next((q for q in parts if re.fullmatch(r"layer_normalization(?:_\d+)?", q)), None)
        if ln is not None and leaf in ("gamma:0", "beta:0"):
            lns.setdefault(ln, {})[leaf] = a
    if not mha and not lns:
        return None
    for sub in ("query", "key", "value", "attention_output"):
        if sub not in mha or "kernel:0" not in mha[sub] or "bias:0" not in mha[sub]:
            raise H5FormatError(f"MultiHeadAttention: {sub} kernel / bias missing")
    f = lambda a: np.ascontiguousarray(a, np.float32)
    att = {"Wq": f(mha["query"]["kernel:0"]), "bq": f(mha["query"]["bias:0"]), "Wk": f(mha["key"]["kernel:0"]), "bk": f(mha["key"]["bias:0"]),
           "Wv": f(mha["value"]["kernel:0"]), "bv": f(mha["value"]["bias:0"]),
           "Wo": f(mha["attention_output"]["kernel:0"]), "bo": f(mha["attention_output"]["bias:0"]), "eps": 1e-3}
    d, heads, dim = att["Wv"].shape if att["Wv"].ndim == 3 else (0, 0, 0)
    if att["Wv"].ndim != 3 or att["bv"].shape != (heads, dim) or att["Wo"].shape != (heads, dim, d) or att["bo"].shape != (d,):
        raise H5FormatError("MultiHeadAttention: value [d, heads, dim] / attention_output [heads, dim, d] expected")
    att["ln"] = []
    for n in sorted(lns, key=lambda s: int(s.rsplit("_", 1)[1]) if s[-1].isdigit() else 0):
        if "gamma:0" not in lns[n] or "beta:0" not in lns[n] or lns[n]["gamma:0"].shape != lns[n]["beta:0"].shape:
            raise H5FormatError(f"layer {n} lacks gamma or beta")
        att["ln"].append((f(lns[n]["gamma:0"]), f(lns[n]["beta:0"])))
    if not att["ln"]:
        raise H5FormatError("densePCA_attention has one LayerNormalization per layer: none found")
    return att


def read_keras_conv_weights(path: str) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Ordered [(kernel[kh,kw,c_in,c_out] f32, bias[c_out] f32)] of the Conv2D layers of a Keras HDF5 file
    (``model.save_weights`` / ``model.save``): layers are the groups named ``conv2d``, ``conv2d_1``, ... and are ordered
    by that index (creation order of the layers), never by their position in the file."""
    ds = read_h5_datasets(path)
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for p, a in ds.items():
        parts = [q for q in p.split("/") if q]
        if parts[-1] not in ("kernel:0", "bias:0"):
            continue
        lname = next((q for q in parts if re.fullmatch(r"conv2d(?:_\d+)?", q)), None)
        if lname is not None:
            layers.setdefault(lname, {})[parts[-1]] = a
    if not layers:
        raise H5FormatError("no conv2d layers found")
    out = []
    for n in sorted(layers, key=lambda s: int(s.split("_")[1]) if "_" in s else 0):
        if "kernel:0" not in layers[n] or "bias:0" not in layers[n]:
            raise H5FormatError(f"layer {n} lacks kernel or bias")
        W = np.ascontiguousarray(layers[n]["kernel:0"], np.float32)
        b = np.ascontiguousarray(layers[n]["bias:0"], np.float32)
        if W.ndim != 4 or W.shape[0] != W.shape[1] or b.shape != (W.shape[3],):
            raise H5FormatError(f"layer {n}: unexpected shapes {W.shape} {b.shape}")
        out.append((W, b))
    return out


def unet_layout_from_weights(weights) -> Tuple[int, Tuple[int, ...], int]:
    """(c_in, widths, c_out) of a UNet-S-shaped Conv2D stack (2 convolutions per encoder level, 2 per decoder level,
    1x1 head), inferred from the kernel shapes; raises if the stack does not have that shape."""
    n = len(weights)
    if n < 7 or (n - 3) % 4 != 0:
        raise ValueError(f"{n} convolutions do not form 2L + 2(L-1) + 1")
    L = (n + 1) // 4
    widths = tuple(int(weights[2 * l][0].shape[3]) for l in range(L))
    c_in, c_out = int(weights[0][0].shape[2]), int(weights[-1][0].shape[3])
    exp = []
    for l in range(L):
        exp += [(3, c_in if l == 0 else widths[l - 1], widths[l]), (3, widths[l], widths[l])]
    for l in range(L - 2, -1, -1):
        exp += [(3, widths[l + 1] + widths[l], widths[l]), (3, widths[l], widths[l])]
    exp.append((1, widths[0], c_out))
    for i, ((W, b), (k, ci, co)) in enumerate(zip(weights, exp)):
        if tuple(W.shape) != (k, k, ci, co):
            raise ValueError(f"convolution {i}: kernel {tuple(W.shape)}, a U-Net of widths {widths} needs {(k, k, ci, co)}")
    return c_in, widths, c_out


def read_maxs(path: str) -> np.ndarray:
    """``np.loadtxt`` of the one-value-per-line ``maxs`` / ``maxs_PCA`` files."""
    return np.atleast_1d(np.loadtxt(path))


def read_scaler_npz(path: str, kind: str):
    """``mean_std.npz`` (kind 'std') or ``min_max_values.npz`` (kind 'min_max')
    -> (in_a, in_b, out_a, out_b) as written by utils.py:299,313."""
    d = np.load(path)
    if kind == "std":
        return d["mean_in"], d["std_in"], d["mean_out"], d["std_out"]
    if kind == "min_max":
        return d["min_in"], d["max_in"], d["min_out"], d["max_out"]
    raise ValueError("Standardization method not valid")


def select_num_pc(explained_variance_ratio: np.ndarray, var: float, max_num_pc: int | None) -> int:
    """PC-count rule.  With ``max_num_pc`` (SM_call.py:86-87): first index whose
    cumulative ratio exceeds ``var`` if that index is in (1, max_num_pc], else
    ``max_num_pc``; without (python_module.py:112-113): the bare argmax."""
    k = int(np.argmax(np.cumsum(explained_variance_ratio) > var))
    if max_num_pc is None:
        return k
    return k if (k > 1 and k <= max_num_pc) else int(max_num_pc)


# --------------------------------------------------------------------------
# padded simulation dataset (utils.read_dataset, utils.py:57-71)
# --------------------------------------------------------------------------
PAD_VALUE = -100.0


def first_index(array, item) -> int:
    """utils.index (utils.py:94-104): index of the first element equal to ``item``.  The reference
    returns None when there is none (and then fails on ``[0]``); here: ``len(array)`` (no padding)."""
    hit = np.flatnonzero(np.asarray(array) == item)
    return int(hit[0]) if hit.size else int(np.shape(array)[0])


class PaddedDataset:
    """Memory-mapped view of the reference's HDF5 dataset file.

    Column order of ``sim_data`` (SM_call.py:386-402): 0 Ux, 1 Uy, 2 p, 3 Cx, 4 Cy, 5-6 delta_U,
    7 delta_p, 8-9 delta_U_prev, 10 delta_p_prev (U_to_gradP / Chapter 4 files carry fewer columns)."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except (ValueError, OSError):
            self._f.close()
            raise H5FormatError("cannot map the dataset file")
        h = _H5(self._mm)
        self._views = {}
        for p, e in h.walk():
            try:
                self._views[p.lstrip("/")] = h.dataset(e["hdr"], copy=False)
            except H5FormatError:
                continue
        for need in ("sim_data", "top_bound", "obst_bound"):
            if need not in self._views:
                self.close()
                raise H5FormatError(f"dataset {need!r} not found (or not contiguous float data)")

    def __getitem__(self, name):
        return self._views[name]

    @property
    def shape(self):
        return self._views["sim_data"].shape

    def read(self, sim: int, time: int):
        """utils.read_dataset(path, sim, time): -> (data[1,1,max_cells,C], top[1,1,m,2], obst[1,1,m,2]) copies."""
        out = []
        for name in ("sim_data", "top_bound", "obst_bound"):
            v = self._views[name]
            if not (0 <= sim < v.shape[0] and 0 <= time < v.shape[1]):
                raise IndexError(f"{name}: sim/time ({sim}, {time}) outside {v.shape[:2]}")
            out.append(np.array(v[sim:sim + 1, time:time + 1, ...]))
        return tuple(out)

    def close(self):
        self._views = {}
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass                      # a caller still holds a view: the mapping goes with it
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_dataset(path: str, sim: int, time: int):
    """Same call as the reference's ``utils.read_dataset`` (utils.py:57-71)."""
    with PaddedDataset(path) as d:
        return d.read(sim, time)


# --------------------------------------------------------------------------
# PCA artefacts
# --------------------------------------------------------------------------
class PCAArtifacts:
    """What the path needs of a fitted (Incremental)PCA: ``transform(X) = (X - mean_) @ components_.T``."""

    def __init__(self, components, mean, explained_variance_ratio):
        self.components_ = np.ascontiguousarray(components, np.float64)
        self.mean_ = np.ascontiguousarray(mean, np.float64)
        self.explained_variance_ratio_ = np.ascontiguousarray(explained_variance_ratio, np.float64)
        if self.components_.ndim != 2 or self.mean_.shape != (self.components_.shape[1],):
            raise ValueError("PCA artefact: components_ [P,K] and mean_ [K] expected")
        if self.explained_variance_ratio_.shape != (self.components_.shape[0],):
            raise ValueError("PCA artefact: explained_variance_ratio_ [P] expected")


def load_pca(path: str) -> PCAArtifacts:
    """``.npz`` export or the reference's pickle (``pk.load(open("ipca_input.pkl","rb"))``, SM_call.py:81).
    A whitened PCA is refused: the path assumes the plain projection."""
    if path.endswith(".npz"):
        d = np.load(path)
        return PCAArtifacts(d["components_"], d["mean_"], d["explained_variance_ratio_"])
    with open(path, "rb") as f:
        obj = pickle.load(f)          # needs the pickled class importable (scikit-learn / dask-ml), like the reference
    if getattr(obj, "whiten", False):
        raise ValueError("whitened PCA objects are not supported")
    return PCAArtifacts(obj.components_, obj.mean_, obj.explained_variance_ratio_)


def save_pca_npz(path: str, pca) -> None:
    np.savez(path, components_=np.asarray(pca.components_), mean_=np.asarray(pca.mean_),
             explained_variance_ratio_=np.asarray(pca.explained_variance_ratio_))


def find_pca(directory: str, stem: str) -> str:
    """``<stem>.pkl`` (reference name) or ``<stem>.npz`` in ``directory``."""
    for ext in (".pkl", ".npz"):
        p = os.path.join(directory, stem + ext)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"{stem}.pkl / {stem}.npz not found in {directory!r}")
