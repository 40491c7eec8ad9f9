"""ctypes front end of oracle/psm_cpu.c (the C / OpenMP restatement of the surrogate path).  TEST INFRASTRUCTURE ONLY:
imported by tests/ and by the cpu_baseline leg of bench.py, never by the package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libpsm_cpu.so")
_VARIANTS = {"chapter5": 0, "deltas": 1, "gradp": 2}
_SCALERS = {"max_abs": 0, "std": 1, "min_max": 2}
_lib = None


class _Model(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("variant", "S", "ov", "c_in", "c_out", "p_in", "p_out", "n_dense", "scaler", "sdf_ch", "strict")] + \
               [(n, C.POINTER(C.c_double)) for n in ("comp_in", "mean_in", "comp_out", "mean_out", "in_a", "in_b", "out_a", "out_b")] + \
               [("W", C.POINTER(C.POINTER(C.c_float))), ("b", C.POINTER(C.POINTER(C.c_float))), ("dims", C.POINTER(C.c_int32)),
                ("out_scale", C.c_double)]


def build():
    """gcc -O3 -mavx2 -mfma -fopenmp (oracle/Makefile)."""
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.psm_cpu_solve_grid.restype = C.c_int
        _lib.psm_cpu_solve_grid.argtypes = [C.POINTER(_Model), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        _lib.psm_cpu_num_blocks.restype = C.c_int
        _lib.psm_cpu_num_blocks.argtypes = [C.c_int] * 5
    return _lib


class CpuModel:
    """Holds the arrays of an ``oracle.psm_oracle.Model`` in the layout psm_cpu.c reads."""

    def __init__(self, om, strict: bool = False):
        if len(getattr(om, "conv1d", ())):
            raise ValueError("psm_cpu.c restates the Dense stacks only (the conv1D_PCA head is covered by the NumPy oracle)")
        if getattr(om, "attention", None):
            raise ValueError("psm_cpu.c restates the Dense stacks only (densePCA_attention is covered by the NumPy oracle)")
        f64 = lambda a, n=None: np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), (n,)) if n else a, np.float64)
        self.keep = [f64(om.comp_in), f64(om.mean_in), f64(om.comp_out), f64(om.mean_out),
                     f64(om.scaler.in_a, om.comp_in.shape[0]), f64(om.scaler.in_b, om.comp_in.shape[0]),
                     f64(om.scaler.out_a, om.comp_out.shape[0]), f64(om.scaler.out_b, om.comp_out.shape[0])]
        self.W = [np.ascontiguousarray(W, np.float32) for W, _ in om.weights]
        self.b = [np.ascontiguousarray(b, np.float32) for _, b in om.weights]
        n = len(self.W)
        self.Wp = (C.POINTER(C.c_float) * n)(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in self.W])
        self.bp = (C.POINTER(C.c_float) * n)(*[b.ctypes.data_as(C.POINTER(C.c_float)) for b in self.b])
        self.dims = np.ascontiguousarray([self.W[0].shape[0]] + [w.shape[1] for w in self.W], np.int32)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self.m = _Model(_VARIANTS[om.variant], om.S, om.overlap(), om.c_in, om.c_out, om.comp_in.shape[0], om.comp_out.shape[0], n,
                        _SCALERS[om.scaler.kind], om.sdf_ch, int(strict), *[p(a) for a in self.keep], self.Wp, self.bp,
                        self.dims.ctypes.data_as(C.POINTER(C.c_int32)), float(om.out_scale))
        self.c_out, self.p_in = om.c_out, om.comp_in.shape[0]


def solve_grid(grid: np.ndarray, cm: CpuModel, threads: int = 0, want_x_input: bool = False):
    """-> fields [Ny, Nx, c_out] float64 (and x_input [B, p_in]); raises where the reference itself raises."""
    lib = load()
    g = np.ascontiguousarray(grid, np.float64)
    Ny, Nx, gc = g.shape
    fields = np.empty((Ny, Nx, cm.c_out))
    B = lib.psm_cpu_num_blocks(cm.m.variant, Ny, Nx, cm.m.S, cm.m.ov)
    x = np.empty((max(B, 1), cm.p_in)) if want_x_input else None
    rc = lib.psm_cpu_solve_grid(C.byref(cm.m), g.ctypes.data_as(C.POINTER(C.c_double)), Ny, Nx, gc,
                                fields.ctypes.data_as(C.POINTER(C.c_double)),
                                x.ctypes.data_as(C.POINTER(C.c_double)) if want_x_input else None, threads)
    if rc < 0:
        raise ValueError("psm_cpu_solve_grid: shape the reference cannot process")
    return (fields, x) if want_x_input else fields
