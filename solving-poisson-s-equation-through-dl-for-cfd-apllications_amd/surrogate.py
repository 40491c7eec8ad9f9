"""Host-side mirror of the reference's surrogate interface on top of the C-ABI.

``GridSurrogate`` owns one ``psm_handle`` (include/psm.h).  The two
``Evaluation*`` classes keep the names, argument order and meaning of the
reference operators for the part of the path that is built so far
(SURVEY.md §8 a7-a12):

* ``pressureSM_deltas.SM_call.Evaluation`` -- ``assemble_prediction`` (SM_call.py:182)
  and the grid-native body of ``timeStep`` (SM_call.py:452-575),
* ``U_to_gradP`` ``Evaluation`` -- ``assemble_prediction`` (Eval_dual_Dense_onlycil.py:255)
  and the grid-native body of ``timeStep`` (:470-547),
* the Chapter-5 solver module -- the grid-native body of ``py_func``
  (python_module.py:299-473) as ``SolverModule.py_func_grid``.

Everything numeric runs in the HIP library; NumPy is only used to hand buffers
over.  Error behaviour follows the reference where it has one (``ValueError(
"Standardization method not valid")``, SM_call.py:525), otherwise ``PsmError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .synthetic import SurrogateModel


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class GridSurrogate:
    """One surrogate model bound to one uniform grid shape on one GPU."""

    def __init__(self, model: SurrogateModel, ny: int, nx: int, max_cases: int = 1, device: int = 0,
                 strict_degenerate: bool = False, precision: str = "f32"):
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be 'f32' or 'bf16'")
        if model.scaler_kind not in _lib.SCALERS:
            raise ValueError("Standardization method not valid")
        self.lib = _lib.load()
        self.model, self.ny, self.nx, self.max_cases = model, int(ny), int(nx), int(max_cases)
        cfg = _lib.psm_config(
            abi_version=_lib.PSM_ABI_VERSION, variant=_lib.VARIANTS[model.variant], block=model.S,
            overlap=0 if model.ov is None else int(model.ov), c_in=model.c_in, c_out=model.c_out,
            p_in=model.p_in, p_out=model.p_out, n_dense=len(model.weights) + (1 if getattr(model, "attention", None) else 0),
            scaler=_lib.SCALERS[model.scaler_kind],
            sdf_channel=model.sdf_ch, device=device, max_cases=max_cases, strict_degenerate=int(strict_degenerate),
            precision=_lib.PRECISIONS[precision])
        h = C.c_void_p()
        _lib.check(self.lib.psm_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._ticket_cases = {}
        try:
            ci, mi, co, mo = _f64(model.comp_in), _f64(model.mean_in), _f64(model.comp_out), _f64(model.mean_out)
            if ci.shape != (model.p_in, model.S ** 2 * model.c_in) or co.shape != (model.p_out, model.S ** 2 * model.c_out):
                raise ValueError("PCA component matrices have the wrong shape")
            self._chk(self.lib.psm_set_pca(h, _p(ci, C.c_double), _p(mi, C.c_double), _p(co, C.c_double), _p(mo, C.c_double)))
            convs = list(getattr(model, "conv1d", None) or [])
            for l, (K, b) in enumerate(convs):                  # conv1D_PCA head: before the Dense layers
                K, b = _f32(K), _f32(b)
                if K.ndim != 3 or b.shape != (K.shape[2],):
                    raise ValueError("Conv1D kernels must be [kernel_size, c_in, c_out] with bias [c_out]")
                self._chk(self.lib.psm_set_conv1d(h, l, len(convs), K.shape[0], K.shape[1], K.shape[2], _p(K, C.c_float), _p(b, C.c_float)))
            att = getattr(model, "attention", None)
            for l, (W, b) in enumerate(model.weights):
                W, b = _f32(W), _f32(b)
                # densePCA_attention: the attention block takes Dense slot 1, the further layers move up by one
                slot = l + 1 if (att and l > 0) else l
                self._chk(self.lib.psm_set_dense(h, slot, W.shape[0], W.shape[1], _p(W, C.c_float), _p(b, C.c_float)))
            if att:
                self._set_attention(att, len(model.weights) - 1)
            ia = _f64(np.broadcast_to(model.in_a, (model.p_in,)))
            ib = _f64(np.broadcast_to(model.in_b, (model.p_in,)))
            oa = _f64(np.broadcast_to(model.out_a, (model.p_out,)))
            ob = _f64(np.broadcast_to(model.out_b, (model.p_out,)))
            self._chk(self.lib.psm_set_scaler(h, _p(ia, C.c_double), _p(ib, C.c_double), _p(oa, C.c_double), _p(ob, C.c_double)))
            self._chk(self.lib.psm_plan_grid(h, self.ny, self.nx))
            self.B = self.lib.psm_num_blocks(h)
        except Exception:
            self.close()
            raise

    def _set_attention(self, att: dict, n_layers: int):
        """The attention part of densePCA_attention (NNs.py:53-64) -> psm_set_attention (slot 1) + one psm_set_layernorm per
        layer; the query / key projections are not passed on: over a sequence of length 1 they cannot change the result."""
        Wv, bv, Wo, bo = _f32(att["Wv"]), _f32(att["bv"]), _f32(att["Wo"]), _f32(att["bo"])
        d, heads, dim = Wv.shape
        if bv.shape != (heads, dim) or Wo.shape != (heads, dim, d) or bo.shape != (d,):
            raise ValueError("attention weights: Wv [d, heads, dim], bv [heads, dim], Wo [heads, dim, d], bo [d]")
        if len(att["ln"]) != n_layers:
            raise ValueError("densePCA_attention has one LayerNormalization per layer")
        self._chk(self.lib.psm_set_attention(self.h, 1, d, heads, dim, _p(Wv, C.c_float), _p(bv, C.c_float), _p(Wo, C.c_float), _p(bo, C.c_float)))
        for i, (g, b) in enumerate(att["ln"]):
            g, b = _f32(g), _f32(b)
            self._chk(self.lib.psm_set_layernorm(self.h, 1 + i, g.shape[0], _p(g, C.c_float), _p(b, C.c_float),
                                                 float(att.get("eps", 1e-3)), 0 if i == 0 else 1))

    # -- plumbing
    def _chk(self, rc):
        return _lib.check(rc, self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.psm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- solves
    def solve(self, grid: np.ndarray, out_scale: Optional[Sequence[float]] = None) -> np.ndarray:
        """grid [Ny,Nx,>=c_in] or [n,Ny,Nx,>=c_in] (normalised, PM:288-297) -> fields [n,Ny,Nx,c_out] f32."""
        g = np.asarray(grid)
        if g.ndim == 3:
            g = g[None]
        if g.ndim != 4 or g.shape[1:3] != (self.ny, self.nx) or g.shape[3] < self.model.c_in:
            raise ValueError(f"grid must be [n,{self.ny},{self.nx},>={self.model.c_in}]")
        g = _f32(g[..., :self.model.c_in])
        n = g.shape[0]
        self._check_bound(g)
        out = np.empty((n, self.ny, self.nx, self.model.c_out), np.float32)
        sc = None
        if out_scale is not None:
            sc = _f32(np.broadcast_to(out_scale, (n,)))
        self._chk(self.lib.psm_solve_grid(self.h, _p(g, C.c_float), n, _p(sc, C.c_float) if sc is not None else None,
                                          _p(out, C.c_float)))
        return out

    def submit(self, grid: np.ndarray, out_scale: Optional[Sequence[float]] = None, out: Optional[np.ndarray] = None) -> int:
        """Asynchronous host-buffer solve (psm_submit_grid_io): returns a ticket; up to PSM_RING_SLOTS (8) in
        flight, each on its own stream (H2D copy, kernels and D2H copy of one ticket are one graph replay).  A pageable
        ``grid`` may be reused on return; a grid / ``out`` array inside a range registered with :meth:`host_register` is
        DMA'd from / into directly and must be left alone until :meth:`wait` returns."""
        g = np.asarray(grid)
        if g.ndim == 3:
            g = g[None]
        if g.ndim != 4 or g.shape[1:3] != (self.ny, self.nx) or g.shape[3] < self.model.c_in:
            raise ValueError(f"grid must be [n,{self.ny},{self.nx},>={self.model.c_in}]")
        g = _f32(g[..., :self.model.c_in])
        n = g.shape[0]
        self._check_bound(g)
        sc = _f32(np.broadcast_to(out_scale, (n,))) if out_scale is not None else None
        if out is not None and (out.dtype != np.float32 or not out.flags.c_contiguous or out.size != n * self.ny * self.nx * self.model.c_out):
            raise ValueError("out must be a contiguous float32 array [n,Ny,Nx,c_out]")
        t = C.c_int64(-1)
        self._chk(self.lib.psm_submit_grid_io(self.h, _p(g, C.c_float), n, _p(sc, C.c_float) if sc is not None else None,
                                              _p(out, C.c_float) if out is not None else None, C.byref(t)))
        self._ticket_cases[t.value] = (n, out, g)               # keeps the arrays alive while the DMA may touch them
        return t.value

    def wait(self, ticket: int) -> np.ndarray:
        """Field(s) of a ticket returned by :meth:`submit` -> [n,Ny,Nx,c_out] f32 (blocks until it has arrived)."""
        rec = self._ticket_cases.get(ticket)
        if rec is None:
            raise ValueError("unknown ticket")
        n, out, _ = rec
        if out is None:
            out = np.empty((n, self.ny, self.nx, self.model.c_out), np.float32)
            self._chk(self.lib.psm_wait_grid(self.h, ticket, _p(out, C.c_float)))
        else:
            self._chk(self.lib.psm_wait_grid(self.h, ticket, None))
            out = out.reshape(n, self.ny, self.nx, self.model.c_out)
        del self._ticket_cases[ticket]
        return out

    # -- zero-copy ring: the caller packs straight into the slot's pinned memory
    def ring_acquire(self):
        """-> (ticket, grid_in [max_cases,Ny,Nx,c_in], fields_out [max_cases,Ny,Nx,c_out]): NumPy views of the next
        slot's pinned buffers (psm_ring_acquire)."""
        t, gi, fo = C.c_int64(-1), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        self._chk(self.lib.psm_ring_acquire(self.h, C.byref(t), C.byref(gi), C.byref(fo)))
        gin = np.ctypeslib.as_array(gi, shape=(self.max_cases, self.ny, self.nx, self.model.c_in))
        fout = np.ctypeslib.as_array(fo, shape=(self.max_cases, self.ny, self.nx, self.model.c_out))
        return t.value, gin, fout

    def ring_submit(self, ticket: int, n_cases: int = 1, out_scale: Optional[Sequence[float]] = None):
        sc = _f32(np.broadcast_to(out_scale, (n_cases,))) if out_scale is not None else None
        self._chk(self.lib.psm_ring_submit(self.h, ticket, n_cases, _p(sc, C.c_float) if sc is not None else None))

    def ring_wait(self, ticket: int):
        self._chk(self.lib.psm_ring_wait(self.h, ticket))

    def ring_release(self, ticket: int):
        """Give an acquired, not yet submitted ticket back (psm_ring_release)."""
        self._chk(self.lib.psm_ring_release(self.h, ticket))

    def host_register(self, arr: np.ndarray):
        """Register a caller-owned contiguous array for direct DMA (psm_host_register); keep it alive until
        :meth:`host_unregister` / close."""
        if not arr.flags.c_contiguous:
            raise ValueError("contiguous array expected")
        self._chk(self.lib.psm_host_register(self.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def host_unregister(self, arr: np.ndarray):
        self._chk(self.lib.psm_host_unregister(self.h, arr.ctypes.data_as(C.c_void_p)))

    def bind_geometry(self, grid, on_device: bool = False, n_cases: int = 1) -> bool:
        """Bind the obstacle geometry (SDF channel of ``grid`` [ny, nx, c_in], or a device pointer with
        ``on_device``; a case batch [n, ny, nx, c_in] binds one geometry per case slot) for the following solves with
        the same number of cases: 6 launches instead of 8 (7 instead of 9 for batches; ``psm_bind_geometry_cases``; the
        reference's computeOnlyOnce / init_func split).  Returns False -- and leaves the solves on the general
        path -- for configurations the fused path does not cover.  CONTRACT (as in psm.h): until unbind_geometry the
        SDF channel of every solved grid keeps the bound flow-cell pattern (``check_bound = True`` verifies it on the
        host-grid entries)."""
        if on_device:
            rc = self.lib.psm_bind_geometry_cases(self.h, C.c_void_p(int(grid)), int(n_cases), 1)
        else:
            g = np.ascontiguousarray(np.asarray(grid)[..., :self.model.c_in], np.float32)
            if g.ndim == 3:
                g = g[None]
            if g.ndim != 4 or g.shape[1:] != (self.ny, self.nx, self.model.c_in):
                raise ValueError(f"grid must be [{self.ny},{self.nx},{self.model.c_in}] or [n,{self.ny},{self.nx},{self.model.c_in}]")
            rc = self.lib.psm_bind_geometry_cases(self.h, g.ctypes.data_as(C.c_void_p), g.shape[0], 0)
        self._bound_mask = None
        if rc == -5:                    # PSM_ERR_UNSUPPORTED: configuration outside the fused path
            return False
        self._chk(rc)
        n = int(n_cases) if on_device else g.shape[0]
        m = np.empty((n, self.ny, self.nx), np.uint8)          # the pattern the library bound (also for device grids)
        self._chk(self.lib.psm_bound_mask(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)), m.size))
        self._bound_mask = m.astype(bool)
        return True

    def unbind_geometry(self):
        self._bound_mask = None
        self._chk(self.lib.psm_unbind_geometry(self.h))

    # Host-grid entries can verify the contract of psm_bind_geometry: with ``check_bound = True`` a grid whose flow-cell
    # pattern differs from the bound one drops the binding.  Off by default (the comparison costs ~30 us per call, more
    # than the binding saves); the evaluators, whose timeStep_grid takes arbitrary grids, switch it on.
    check_bound = False

    def _check_bound(self, g: np.ndarray):
        if not self.check_bound:
            return
        m = getattr(self, "_bound_mask", None)
        if m is not None and g.shape[0] == m.shape[0] and not np.array_equal(g[..., self.model.sdf_ch] != 0, m):
            self.unbind_geometry()

    @property
    def geometry_bound(self) -> bool:
        return bool(self.lib.psm_geometry_bound(self.h))

    @property
    def guard_trips(self) -> int:
        """Solves so far whose grid was not the bound geometry (detected on the device; each dropped the binding)."""
        return int(self.lib.psm_guard_trips(self.h))

    def solve_device(self, d_grid: int, n_cases: int, d_fields: int, stream: int = 0,
                     out_scale: Optional[Sequence[float]] = None):
        """Asynchronous solve on raw device pointers (e.g. ``torch.Tensor.data_ptr()``)."""
        sc = None
        if out_scale is not None:
            sc = _f32(np.broadcast_to(out_scale, (n_cases,)))
        self._chk(self.lib.psm_solve_grid_device(self.h, C.c_void_p(d_grid), n_cases,
                                                 _p(sc, C.c_float) if sc is not None else None,
                                                 C.c_void_p(d_fields), C.c_void_p(stream)))

    def synchronize(self):
        self._chk(self.lib.psm_synchronize(self.h))

    def reassemble(self, grid: np.ndarray, block_pred: np.ndarray) -> np.ndarray:
        """Reassembly alone: block_pred [B,S,S,c_out] (or [B,S,S] when c_out==1) -> [Ny,Nx,c_out]."""
        g = _f32(np.asarray(grid)[..., :self.model.c_in])
        bp = _f32(np.asarray(block_pred).reshape(self.B, -1))
        if bp.shape[1] != self.model.S ** 2 * self.model.c_out:
            raise ValueError("block_pred has the wrong shape")
        out = np.empty((self.ny, self.nx, self.model.c_out), np.float32)
        self._chk(self.lib.psm_reassemble(self.h, _p(g, C.c_float), _p(bp, C.c_float), _p(out, C.c_float)))
        return out

    def label_blocks(self, grid: np.ndarray, labels: np.ndarray) -> np.ndarray:
        """Label blocks of the planned layout with the per-block flow-cell mean removed (SM_call.py:487-488;
        Eval_dual_Dense_onlycil.py:509-511): labels [Ny,Nx,c_out] (or [Ny,Nx]) -> [B,S,S,c_out] float32."""
        g = _f32(np.asarray(grid)[..., :self.model.c_in])
        lab = _f32(np.asarray(labels).reshape(self.ny, self.nx, self.model.c_out))
        if g.shape != (self.ny, self.nx, self.model.c_in):
            raise ValueError("grid has the wrong shape")
        out = np.empty((self.B, self.model.S, self.model.S, self.model.c_out), np.float32)
        self._chk(self.lib.psm_label_blocks(self.h, _p(g, C.c_float), _p(lab, C.c_float), _p(out, C.c_float)))
        return out

    def block_error(self, grid: np.ndarray, labels: np.ndarray) -> dict:
        """``utils.compute_in_block_error`` (pressureSM_deltas/utils.py:210-243) for the LAST solve: decoded blocks against the
        de-meaned label blocks over the flow cells, before the reassembly.  ``labels`` [Ny,Nx,c_out] (or [Ny,Nx]) in the
        network's normalised output units (scaled by the solve's out_scale like SM_call.py:555).  -> the two values the
        reference appends (``mean_err``, ``mean_sq_err``) and the printed normVal / biasNorm / stdeNorm / rmseNorm."""
        g = _f32(np.asarray(grid)[..., :self.model.c_in])
        lab = _f32(np.asarray(labels).reshape(self.ny, self.nx, self.model.c_out))
        if g.shape != (self.ny, self.nx, self.model.c_in):
            raise ValueError("grid has the wrong shape")
        out = (C.c_double * 5)()
        self._chk(self.lib.psm_block_error(self.h, _p(g, C.c_float), _p(lab, C.c_float), out))
        bias, rmse = out[0] * 100, np.sqrt(out[1]) * 100
        with np.errstate(invalid="ignore"):
            stde = np.sqrt(rmse ** 2 - bias ** 2)
        return {"mean_err": float(out[0]), "mean_sq_err": float(out[1]), "normVal": float(out[2]), "norm_pred": float(out[3]),
                "n": int(out[4]), "biasNorm": float(bias), "rmseNorm": float(rmse), "stdeNorm": float(stde)}

    def gaussian_filter(self, field: np.ndarray, sigma=(10.0, 10.0)) -> np.ndarray:
        """scipy.ndimage.gaussian_filter(field, sigma, order=0) on the GPU (SM_call.py:353-363)."""
        f = _f32(field)
        if f.ndim != 2:
            raise ValueError("field must be 2-D")
        out = np.empty_like(f)
        self._chk(self.lib.psm_gaussian_filter(self.h, _p(f, C.c_float), f.shape[0], f.shape[1], float(sigma[0]),
                                                float(sigma[1]), _p(out, C.c_float)))
        return out

    def poisson_features(self, ux, uy, dux, duy, sdfunct, L, U, k, max_abs) -> np.ndarray:
        """pressureSM_Poisson input image (SM_call.py:588-711): float64 grids [Ny,Nx] -> grid [Ny,Nx,4] float32."""
        arrs = [_f64(np.asarray(a)) for a in (ux, uy, dux, duy, sdfunct)]
        ny, nx = arrs[0].shape
        if any(a.shape != (ny, nx) for a in arrs):
            raise ValueError("ux, uy, dux, duy, sdfunct must share one [Ny,Nx] shape")
        params = _f64(np.array([L, U, k, *max_abs], np.float64))
        if params.shape != (7,):
            raise ValueError("max_abs must hold 4 scales")
        out = np.empty((ny, nx, 4), np.float32)
        self._chk(self.lib.psm_poisson_features(self.h, *[_p(a, C.c_double) for a in arrs], ny, nx,
                                                _p(params, C.c_double), _p(out, C.c_float)))
        return out

    def set_integration(self, sdfunct: np.ndarray, center_y: int, center_x: int, dx: float, dy: float):
        """Geometry of the gradP -> p integration (Eval_dual_Dense_onlycil.py:592-628)."""
        sd = _f64(sdfunct)
        self._integ_shape = sd.shape
        self._chk(self.lib.psm_set_integration(self.h, sd.shape[0], sd.shape[1], _p(sd, C.c_double), int(center_y),
                                                int(center_x), float(dx), float(dy)))

    def integrate_gradp(self, gradp: np.ndarray) -> np.ndarray:
        """(dp/dx, dp/dy) [Ny,Nx,2] -> p [Ny,Nx] (integrate_field + four-quadrant stitching)."""
        g = _f32(gradp)
        if g.shape != tuple(self._integ_shape) + (2,):
            raise ValueError("gradp must be [Ny,Nx,2] of the integration geometry")
        out = np.empty(self._integ_shape, np.float32)
        self._chk(self.lib.psm_integrate_gradp(self.h, _p(g, C.c_float), _p(out, C.c_float)))
        return out

    # -- introspection
    def stage(self, name: str, n_cases: int = 1) -> np.ndarray:
        m = self.model
        rows = n_cases * self.B
        shape = {"x_input": (rows, m.p_in), "res": (rows, m.p_out), "block_pred": (rows, m.S, m.S, m.c_out),
                 "offsets": (n_cases, m.c_out, self.B), "shift": (n_cases, m.c_out)}[name]
        out = np.empty(shape, np.float32)
        self._chk(self.lib.psm_read_stage(self.h, _lib.STAGES[name], _p(out, C.c_float), out.size))
        return out

    def profile(self, d_grid: int, n_cases: int, d_fields: int) -> dict:
        ms = (C.c_float * len(_lib.KERNELS))()
        self._chk(self.lib.psm_profile_solve(self.h, C.c_void_p(d_grid), n_cases, C.c_void_p(d_fields), ms))
        return dict(zip(_lib.KERNELS, [float(v) for v in ms]))

    def enable_kernel_timing(self, kernel: str, on=True, repeat: int = 1):
        """Event-time one kernel group; ``repeat`` launches per event pair (see psm.h)."""
        self._chk(self.lib.psm_enable_kernel_timing(self.h, _lib.KERNELS.index(kernel), int(repeat) if on else 0))

    def event_pair_overhead_ms(self, n: int = 200) -> float:
        t = C.c_double()
        self._chk(self.lib.psm_event_pair_overhead(self.h, n, C.byref(t)))
        return t.value

    def kernel_timing(self, kernel: str):
        t, n = C.c_double(), C.c_int64()
        self._chk(self.lib.psm_get_kernel_timing(self.h, _lib.KERNELS.index(kernel), C.byref(t), C.byref(n)))
        return t.value, n.value


# ---------------------------------------------------------------------------
# host-only helpers (no GPU): layout / owner map / host replay of the reassembly
# ---------------------------------------------------------------------------
def layout(variant: str, ny: int, nx: int, S: int = 128, ov: int = 0):
    """-> (blocks[B,4] = (y0, x0, idx_i, idx_j), n_x, n_y) as enumerated at
    PM:306-329 / SMD:461-479 / UGP:479-500."""
    lib = _lib.load()
    nxv, nyv = C.c_int32(), C.c_int32()
    n = lib.psm_layout(_lib.VARIANTS[variant], ny, nx, S, ov, None, 0, C.byref(nxv), C.byref(nyv))
    if n < 0:
        raise _lib.PsmError(n, _lib.last_error())
    blocks = np.zeros((n, 4), np.int32)
    lib.psm_layout(_lib.VARIANTS[variant], ny, nx, S, ov, _p(blocks, C.c_int32), n, None, None)
    return blocks, nxv.value, nyv.value


def owner_map(variant: str, ny: int, nx: int, S: int = 128, ov: int = 0, strict: bool = False) -> np.ndarray:
    out = np.empty((ny, nx), np.int32)
    _lib.check(_lib.load().psm_owner_map(_lib.VARIANTS[variant], ny, nx, S, ov, int(strict), _p(out, C.c_int32)))
    return out


def debug_reassemble_host(variant: str, grid: np.ndarray, block_pred: np.ndarray, c_out: int, S: int = 128,
                          ov: int = 0, strict: bool = False, sdf_ch: int = 2):
    """Host replay of the device reassembly tables (verification helper)."""
    g = _f32(grid)
    ny, nx, c_in = g.shape
    B = block_pred.shape[0]
    bp = _f32(np.asarray(block_pred).reshape(B, -1))
    fields = np.empty((ny, nx, c_out), np.float32)
    offs = np.empty((c_out, B), np.float32)
    sh = np.empty((c_out,), np.float32)
    _lib.check(_lib.load().psm_debug_reassemble_host(
        _lib.VARIANTS[variant], ny, nx, S, ov, int(strict), c_in, c_out, sdf_ch, _p(g, C.c_float), _p(bp, C.c_float),
        _p(fields, C.c_float), _p(offs, C.c_float), _p(sh, C.c_float)))
    return fields, offs, sh


# ---------------------------------------------------------------------------
# reference-shaped operators
# ---------------------------------------------------------------------------
def _grid_from_blocks(x_array: np.ndarray, blocks: np.ndarray, ny: int, nx: int) -> np.ndarray:
    """Rebuild the grid image from the overlapping input blocks (the reference keeps
    only ``self.x_array`` around when it calls ``assemble_prediction``)."""
    S = x_array.shape[1]
    g = np.zeros((ny, nx, x_array.shape[3]), np.float32)
    for b, (y0, x0, _, _) in enumerate(blocks):
        g[y0:y0 + S, x0:x0 + S] = x_array[b]
    return g


def load_artifacts(variant, model_path, directory, var_p, var_in, max_num_PC, standardization_method, shape, overlap,
                   c_in: int = 3, sdf_ch: int = 2):
    """What the reference's ``Evaluation.__init__`` loads (SM_call.py:70-87; Eval_dual_Dense_onlycil.py:51-66):
    ``maxs`` (+ ``maxs_PCA`` for 'max_abs'), the Keras network of ``model_path`` (Dense stack, HDF5), the two
    pickled PCA objects (``ipca_input``, ``ipca_p``: ``.pkl`` like the reference or the ``.npz`` export) with the
    component-count rule of :86-87, and the scaler file of the chosen standardisation (:507-519).
    -> (SurrogateModel, maxs)."""
    from . import formats
    maxs = formats.read_maxs(os.path.join(directory, "maxs"))
    convs, weights = formats.read_keras_conv1d_head(model_path)     # Dense stack, or the conv1D_PCA head (NNs.py:75-124)
    attention = formats.read_keras_attention(model_path)            # densePCA_attention (NNs.py:40-72), else None
    pin = formats.load_pca(formats.find_pca(directory, "ipca_input"))
    pout = formats.load_pca(formats.find_pca(directory, "ipca_p"))
    pc_p = formats.select_num_pc(pout.explained_variance_ratio_, var_p, max_num_PC)
    pc_in = formats.select_num_pc(pin.explained_variance_ratio_, var_in, max_num_PC)
    c_out = 2 if variant == "gradp" else 1
    if pin.components_.shape[1] != shape * shape * c_in or pout.components_.shape[1] != shape * shape * c_out:
        raise ValueError("PCA artefacts do not match the block shape / channel count")
    first_in = weights[0][0].shape[0] // (convs[-1][0].shape[2] if convs else 1)
    if first_in != pc_in or weights[-1][0].shape[1] != pc_p:
        raise ValueError(f"network is {first_in} -> {weights[-1][0].shape[1]} but the PCA rule gives "
                         f"{pc_in} -> {pc_p} components")
    m = SurrogateModel(variant, c_in, c_out, pin.components_[:pc_in], pin.mean_, pout.components_[:pc_p], pout.mean_,
                       list(weights), scaler_kind=standardization_method, S=shape)
    m.conv1d = list(convs)
    m.attention = attention
    m.ov = int(overlap) if overlap else None                 # deltas: overlap in cells; gradp: `avance`
    m.sdf_ch = sdf_ch
    if standardization_method == "max_abs":
        mp = formats.read_maxs(os.path.join(directory, "maxs_PCA"))      # Eval_dual_Dense_onlycil.py:52-55
        m.in_a, m.out_a = float(mp[0]), float(mp[1])
    else:
        fn = "mean_std.npz" if standardization_method == "std" else "min_max_values.npz"
        a, b, c, d = formats.read_scaler_npz(os.path.join(directory, fn), standardization_method)
        m.in_a, m.in_b, m.out_a, m.out_b = (np.asarray(a, np.float64)[:pc_in], np.asarray(b, np.float64)[:pc_in],
                                            np.asarray(c, np.float64)[:pc_p], np.asarray(d, np.float64)[:pc_p])
    return m, maxs


def error_metrics(pred, truth, no_flow_bool) -> dict:
    """The error summary every evaluator of the reference prints per frame (SM_call.py:696-724;
    pressureSM_Poisson/SM_call.py:962-994; Eval_dual_Dense_onlycil.py:667-687): over the flow cells, with
    norm = max - min of the truth there and NaN differences left out,
    BIAS = mean(pred - true) / norm, RMSE = sqrt(mean((pred - true)^2)) / norm, STDE = sqrt(RMSE^2 - BIAS^2), in percent;
    ``mean_err`` / ``mean_sq_err`` are the two values the reference appends to ``pred_minus_true`` / ``pred_minus_true_squared``."""
    true_masked, pred_masked = np.asarray(truth)[~no_flow_bool], np.asarray(pred)[~no_flow_bool]
    norm = np.max(true_masked) - np.min(true_masked)
    diff = pred_masked - true_masked
    diff = diff[~np.isnan(diff)]
    bias = np.mean(diff) / norm * 100
    rmse = np.sqrt(np.mean(diff ** 2)) / norm * 100
    with np.errstate(invalid="ignore"):
        stde = np.sqrt(rmse ** 2 - bias ** 2)
    return {"normVal": float(norm), "biasNorm": float(bias), "stdeNorm": float(stde), "rmseNorm": float(rmse),
            "mean_err": float(np.mean(diff) / norm), "mean_sq_err": float(np.mean(diff ** 2) / norm ** 2)}


def _summary(b, s) -> dict:
    """BIAS / RMSE / STDE [%] of a list of frames as the reference's mains print them (SM_call.py:887-895)."""
    bias, rmse = np.mean(b) * 100, np.sqrt(np.mean(s)) * 100
    return {"BIAS": float(bias), "RMSE": float(rmse), "STDE": float(np.sqrt(max(rmse ** 2 - bias ** 2, 0.0)))}


class Evaluation:
    """``pressureSM_deltas.SM_call.Evaluation`` (SM_call.py:26-87) on the GPU path.

    ``model`` carries the artefacts the reference loads from the working
    directory (``maxs``, the Keras model, ``ipca_*.pkl``, ``mean_std.npz``)."""
    variant = "deltas"

    c_in_expected, sdf_ch_expected = 3, 2

    def __init__(self, delta, shape, overlap, var_p, var_in, dataset_path, model_path, max_num_PC,
                 standardization_method, model: SurrogateModel = None, device: int = 0, artifact_dir: str = None):
        if standardization_method not in ("std", "min_max", "max_abs"):
            raise ValueError("Standardization method not valid")
        self.maxs = None
        if model is None:
            # like the reference: `maxs`, `ipca_input.pkl`, `ipca_p.pkl` and the scaler files are read from
            # the working directory (SM_call.py:70-87, 507-519), the network from model_path (:74-79)
            model, self.maxs = load_artifacts(self.variant, model_path, artifact_dir or os.getcwd(), var_p, var_in,
                                              max_num_PC, standardization_method, shape, overlap, self.c_in_expected,
                                              self.sdf_ch_expected)
        self.delta, self.shape, self.overlap = delta, shape, overlap
        self.var_p, self.var_in, self.dataset_path, self.max_num_PC = var_p, var_in, dataset_path, max_num_PC
        self.standardization_method = standardization_method
        self.artifacts = model
        self.pc_in, self.pc_p = model.p_in, model.p_out
        self.device = device
        self.Ref_BC = 0
        self._sur = None
        self.x_array = None

    def _surrogate(self, ny, nx):
        if self._sur is None or (self._sur.ny, self._sur.nx) != (ny, nx):
            if self._sur is not None:
                self._sur.close()
            self._sur = GridSurrogate(self.artifacts, ny, nx, 1, self.device)
            self._sur.check_bound = True          # timeStep_grid takes arbitrary grids: a foreign geometry drops the binding
        return self._sur

    # ---- dataset-driven entry points (same names and arguments as the reference) -------------------
    def computeOnlyOnce(self, sim):
        """SM_call.py:89-180: geometry of simulation ``sim`` (frame 0 of the dataset) -> interpolation tables,
        SDF image, index map; handed to the GPU library (psm_set_geometry).  Returns 0."""
        from . import formats
        from .geometry import build_geometry_evaluator
        data, top_b, obst_b = formats.read_dataset(self.dataset_path, sim, 0)
        self.indice = formats.first_index(data[0, 0, :, 0], formats.PAD_VALUE)
        cells = np.asarray(data[0, 0, :self.indice], np.float64)
        top = np.asarray(top_b[0, 0, :formats.first_index(top_b[0, 0, :, 0], formats.PAD_VALUE)], np.float64)
        obst = np.asarray(obst_b[0, 0, :formats.first_index(obst_b[0, 0, :, 0], formats.PAD_VALUE)], np.float64)
        t = build_geometry_evaluator(cells[:, 3:5], cells[:, 2], top, obst, self.delta, idw_fallback=True)   # utils.interp_weights
        self.grid_shape_y, self.grid_shape_x = t.ny, t.nx
        self.vert, self.weights, self.indices, self.sdfunct = t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct[:, :, None]
        sur = self._surrogate(t.ny, t.nx)
        v1, w1 = np.ascontiguousarray(t.vtx_m2g, np.int32), _f64(t.wts_m2g)
        idx, sdf = np.ascontiguousarray(t.indices, np.int32), _f64(t.sdfunct)
        mx = _f64(np.asarray(self.maxs if self.maxs is not None else (1.0, 1.0, 1.0, 1.0), np.float64)[:4])
        sur._chk(sur.lib.psm_set_geometry(sur.h, int(self.indice), t.ny, t.nx, _p(v1, C.c_int32), _p(w1, C.c_double),
                                          _p(idx, C.c_int32), _p(sdf, C.c_double), None, None, _p(mx, C.c_double), 1, 1, 0.05))
        self.tables = t
        self._bind_simulation_geometry(sur, t.sdfunct, float(mx[2]))
        return 0

    def _bind_simulation_geometry(self, sur, sdfunct: np.ndarray, max_abs_dist: float):
        """computeOnlyOnce fixes the obstacle for every timeStep of the simulation: bind its flow-cell pattern (the SDF
        channel exactly as timeStep normalises it) so that the steps take the 6-launch path.  Grids of another
        geometry passed to timeStep_grid drop the binding again (GridSurrogate._check_bound)."""
        g = np.zeros((sur.ny, sur.nx, self.artifacts.c_in), np.float32)
        sd = np.asarray(sdfunct, np.float64) / max_abs_dist
        g[..., self.artifacts.sdf_ch] = np.where(np.isnan(sd), 0.0, sd).astype(np.float32)
        sur.bind_geometry(g)

    def _mesh_to_grid(self, columns: np.ndarray) -> np.ndarray:
        """interpolate_fill + scatter of k cell columns on the GPU (psm_mesh_to_grid) -> [Ny,Nx,k] float64."""
        sur = self._surrogate(self.grid_shape_y, self.grid_shape_x)
        v = _f64(columns)
        out = np.empty((self.grid_shape_y, self.grid_shape_x, v.shape[1]), np.float64)
        sur._chk(sur.lib.psm_mesh_to_grid(sur.h, _p(v, C.c_double), v.shape[0], v.shape[1], 1, _p(out, C.c_double)))
        return out

    def timeStep(self, sim, time, plot_intermediate_fields=False, save_plots=False, show_plots=False, apply_filter=False):
        """SM_call.py:367-575 without the plots and error prints: frame (sim, time) of the dataset -> assembled
        delta-p image [Ny,Nx] (dimensional, like ``deltap_res``), or 0 for an irrelevant time step (:417-421).
        Also kept: ``self.cfd_results`` (the frame's own delta-p image) and ``self.no_flow_bool``."""
        from . import formats
        if getattr(self, "tables", None) is None:
            raise RuntimeError("computeOnlyOnce has not been called")
        data, _, _ = formats.read_dataset(self.dataset_path, sim, time)
        d = data[0, 0, :self.indice]                  # float32 like the file: the reference normalises in float32
        Ux, Uy, p = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        delta_U, delta_p = d[:, 5:7], d[:, 7:8]
        delta_U_prev, delta_p_prev = d[:, 8:10], d[:, 10:11]
        deltaU_changed = np.abs(delta_U - delta_U_prev).sum(axis=-1)                     # :400-401
        deltaU_changed = deltaU_changed / deltaU_changed.max()
        U_max_norm = np.max(np.sqrt(np.square(Ux) + np.square(Uy)))                      # :407
        deltaU_max_norm = np.max(np.sqrt(np.square(delta_U[:, 0:1]) + np.square(delta_U[:, 1:2])))
        if (deltaU_max_norm / U_max_norm) < 1e-4:                                       # :413-421
            return 0
        cols = np.concatenate([delta_U / U_max_norm, delta_p / pow(U_max_norm, 2.0), p, deltaU_changed[:, None], delta_p_prev],
                              axis=1).astype(np.float64)
        g = self._mesh_to_grid(cols)                                                    # :423-431
        max_abs_Ux, max_abs_Uy, max_abs_dist, max_abs_p = [float(v) for v in self.maxs[:4]]
        grid = np.zeros((self.grid_shape_y, self.grid_shape_x, 5))
        grid[..., 0:2] = g[..., 0:2]
        grid[..., 2] = self.sdfunct[..., 0]
        grid[..., 3:5] = g[..., 2:4]
        grid[np.isnan(grid)] = 0                                                        # :439
        grid[..., 0] /= max_abs_Ux; grid[..., 1] /= max_abs_Uy                          # :442-445
        grid[..., 2] /= max_abs_dist; grid[..., 3] /= max_abs_p
        self.grid = grid
        self.deltaU_change_grid, self.deltaP_prev_grid = g[..., 4], g[..., 5]            # :448-451 (NaNs kept)
        self.max_abs_p = max_abs_p
        res = self.timeStep_grid(grid[..., :3], U_max_norm, max_abs_p)                   # :452-572
        # :553-557 the error of the decoded blocks against the de-meaned label blocks, before the assembly
        mb = self._surrogate(*grid.shape[:2]).block_error(grid[..., :3], grid[..., 3])
        for name, key in (("pred_minus_true_block", "mean_err"), ("pred_minus_true_squared_block", "mean_sq_err")):
            if not hasattr(self, name):
                setattr(self, name, [])
            getattr(self, name).append(mb[key])
        if not isinstance(getattr(self, "last_metrics", None), dict):
            self.last_metrics = {}
        self.last_metrics["blocks"] = mb
        if apply_filter:
            res = self._surrogate(*grid.shape[:2]).gaussian_filter(res, (10, 10))
        self.cfd_results = grid[..., 3] * max_abs_p * pow(U_max_norm, 2.0)               # :580 (float32 square, like the reference)
        self.no_flow_bool = grid[..., 2] == 0
        self.U_max_norm = float(U_max_norm)
        self._record_errors(res, self.cfd_results, self.no_flow_bool)
        return res

    print_metrics = False           # True: print the per-frame error block like the reference's timeStep does

    def _record_errors(self, field, truth, no_flow_bool, suffix: str = "", title: str = None):
        """SM_call.py:696-724: normalised bias / squared error of the assembled field over the flow cells, appended to
        ``pred_minus_true<suffix>`` / ``pred_minus_true_squared<suffix>`` (what the mains average); the frame's
        normVal / biasNorm / stdeNorm / rmseNorm are kept in ``self.last_metrics[suffix or 'delta_p']``."""
        m = error_metrics(field, truth, no_flow_bool)
        for name in ("pred_minus_true" + suffix, "pred_minus_true_squared" + suffix):
            if not hasattr(self, name):
                setattr(self, name, [])
        getattr(self, "pred_minus_true" + suffix).append(m["mean_err"])
        getattr(self, "pred_minus_true_squared" + suffix).append(m["mean_sq_err"])
        if not isinstance(getattr(self, "last_metrics", None), dict):
            self.last_metrics = {}
        self.last_metrics[suffix.lstrip("_") or "delta_p"] = m
        if self.print_metrics:
            print(f"""
		{'** ' + title + ' **' if title else ''}
		normVal  = {m['normVal']} Pa
		biasNorm = {m['biasNorm']:.3f}%
		stdeNorm = {m['stdeNorm']:.3f}%
		rmseNorm = {m['rmseNorm']:.3f}%
		""", flush=True)
        return m

    def timeStep_grid(self, grid: np.ndarray, U_max_norm: float = 1.0, max_abs_p: float = 1.0) -> np.ndarray:
        """Grid-native body of ``timeStep`` (SM_call.py:452-575): -> deltap_res [Ny,Nx]."""
        sur = self._surrogate(grid.shape[0], grid.shape[1])
        return sur.solve(grid, out_scale=[max_abs_p * U_max_norm ** 2])[0, :, :, 0]

    def label_self_check(self, grid: np.ndarray, labels: np.ndarray) -> np.ndarray:
        """The reference's own check of the assembly algorithm: the CFD labels, de-meaned per block over the flow cells
        (SM_call.py:487-488; Eval_dual_Dense_onlycil.py:509-511), pushed through the same reassembly as the prediction
        ("it should be almost perfect in that case", SM_call.py:577-580; live as test_dPdx / test_dPdy at
        Eval_dual_Dense_onlycil.py:546-547).  ``grid`` [Ny,Nx,>=c_in] normalised input image (flow mask = its SDF
        channel), ``labels`` [Ny,Nx,c_out] -> assembled label field(s) [Ny,Nx,c_out]; also keeps ``self.y_array``."""
        sur = self._surrogate(grid.shape[0], grid.shape[1])
        self.y_array = sur.label_blocks(grid, labels)
        return sur.reassemble(grid, self.y_array)

    def assemble_prediction(self, array, indices_list, n_x, n_y, apply_filter, shape_x, shape_y,
                            deltaU_change_grid=None, deltaP_prev_grid=None, apply_deltaU_change_wgt=False):
        """SM_call.py:182: ``array`` [B,S,S] corrected and pasted into [shape_y, shape_x].
        The flow mask comes from ``self.x_array`` like in the reference."""
        if self.x_array is None:
            raise ValueError("self.x_array must hold the input blocks")
        blocks, nx_, ny_ = layout(self.variant, shape_y, shape_x, self.shape, self.overlap)
        if (nx_, ny_) != (n_x, n_y) or len(indices_list) != len(blocks):
            raise ValueError("n_x / n_y / indices_list do not match this grid")
        sur = self._surrogate(shape_y, shape_x)
        grid = _grid_from_blocks(np.asarray(self.x_array, np.float32), blocks, shape_y, shape_x)
        result = sur.reassemble(grid, np.asarray(array))[..., 0]
        return self._post_steps(sur, result, apply_filter, deltaU_change_grid, deltaP_prev_grid, apply_deltaU_change_wgt)

    @staticmethod
    def _post_steps(sur, result, apply_filter, deltaU_change_grid, deltaP_prev_grid, apply_deltaU_change_wgt):
        """Tail of ``assemble_prediction`` (SM_call.py:352-363): optional Gaussian filter and deltaU-change weighting."""
        filter_tuple = (10, 10)                                   # SM_call.py:353
        if apply_filter:
            result = sur.gaussian_filter(result, filter_tuple)
        change_in_deltap = None
        if apply_deltaU_change_wgt:                               # SM_call.py:359-363
            w = sur.gaussian_filter(np.asarray(deltaU_change_grid, np.float32), (50, 50))
            change_in_deltap = (result - np.asarray(deltaP_prev_grid, np.float32)) * w
            change_in_deltap = sur.gaussian_filter(change_in_deltap, filter_tuple)
        return result, change_in_deltap


def call_SM_main(delta, model_name, shape, overlap_ratio, var_p, var_in, max_num_PC, dataset_path,
                 plot_intermediate_fields=False, standardization_method="std", save_plots=False, show_plots=False,
                 apply_filter=False, create_GIF=False, n_sims=1, n_ts=1, device: int = 0, artifact_dir: str = None):
    """``pressureSM_deltas.SM_call.call_SM_main`` (SM_call.py:778-900): evaluate ``n_ts`` frames of ``n_sims``
    simulations of the dataset and return the error summary the reference prints -- per simulation and overall
    BIAS / STDE / RMSE [%] of delta-p over the flow cells, and BIAS_block / RSME_block / STDE_block [%] of the decoded blocks
    before the assembly (SM_call.py:824-826, from ``utils.compute_in_block_error``); plots and GIFs are not produced."""
    overlap = int(overlap_ratio * shape)
    ev = Evaluation(delta, shape, overlap, var_p, var_in, dataset_path, model_name, max_num_PC, standardization_method,
                    device=device, artifact_dir=artifact_dir)
    ev.pred_minus_true, ev.pred_minus_true_squared = [], []
    ev.pred_minus_true_block, ev.pred_minus_true_squared_block = [], []

    summary = _summary
    out = {"sims": []}
    for sim in range(n_sims):
        n0 = len(ev.pred_minus_true)
        ev.computeOnlyOnce(sim)
        for time in range(n_ts):
            ev.timeStep(sim, time, plot_intermediate_fields, save_plots, show_plots, apply_filter)
        if len(ev.pred_minus_true) > n0:
            s = summary(ev.pred_minus_true[n0:], ev.pred_minus_true_squared[n0:])
            # SM_call.py:824-826 BIAS_block / RSME_block / STDE_block: the same summary over the per-frame block errors
            blk = summary(ev.pred_minus_true_block[n0:], ev.pred_minus_true_squared_block[n0:])
            s.update(BIAS_block=blk["BIAS"], RSME_block=blk["RMSE"], STDE_block=blk["STDE"])
            out["sims"].append(s)
        else:
            out["sims"].append(None)                       # every frame of this simulation was irrelevant
    if ev.pred_minus_true:
        out["overall"] = summary(ev.pred_minus_true, ev.pred_minus_true_squared)
        blk = summary(ev.pred_minus_true_block, ev.pred_minus_true_squared_block)
        out["overall"].update(BIAS_block=blk["BIAS"], RSME_block=blk["RMSE"], STDE_block=blk["STDE"])
    return out


def call_SM_main_Poisson(delta, model_name, shape, overlap_ratio, var_p, var_in, max_num_PC, dataset_path,
                         plot_intermediate_fields, standardization_method, k, save_plots, show_plots, apply_filter, create_GIF,
                         n_sims, n_ts, phis_fn, device: int = 0, artifact_dir: str = None, sim_offset: int = 1, time_offset: int = 16):
    """``pressureSM_Poisson.SM_call.call_SM_main`` (pressureSM_Poisson/SM_call.py:1069-1170), same argument list: evaluates
    frames ``time_offset .. time_offset + n_ts`` of simulations ``sim_offset .. sim_offset + n_sims`` (the reference's
    ``sim += 1`` / ``time += 16``, :1095, :1104) with ``phi = phi_list[sim]`` from ``phis_fn`` and returns the three error
    summaries it prints -- delta-p with the deltaU-change weighting, delta-p without it, and p -- per simulation (last
    ``n_ts`` frames, :1110-1116) and overall.  Plots and GIFs are not produced."""
    overlap = int(overlap_ratio * shape)
    ev = EvaluationPoisson(delta, shape, overlap, var_p, var_in, dataset_path, model_name, max_num_PC, standardization_method,
                           k, phis_fn, device=device, artifact_dir=artifact_dir)
    for sfx in ("", "_deltap_crude", "_p"):
        setattr(ev, "pred_minus_true" + sfx, [])
        setattr(ev, "pred_minus_true_squared" + sfx, [])
    phi_list = np.atleast_1d(np.loadtxt(phis_fn, dtype=float))
    out = {"sims": []}
    for sim in range(n_sims):
        sim += sim_offset
        ev.computeOnlyOnce(sim)
        phi = phi_list[sim]
        n0 = len(ev.pred_minus_true)
        for time in range(n_ts):
            ev.timeStep(sim, time + time_offset, plot_intermediate_fields, save_plots, show_plots, apply_filter, phi)
        if len(ev.pred_minus_true) > n0:
            out["sims"].append({"sim": sim, "phi": float(phi),
                                "delta_p": _summary(ev.pred_minus_true[-n_ts:], ev.pred_minus_true_squared[-n_ts:]),
                                "delta_p_no_weighting": _summary(ev.pred_minus_true_deltap_crude[-n_ts:], ev.pred_minus_true_squared_deltap_crude[-n_ts:])})
        else:
            out["sims"].append(None)
    if ev.pred_minus_true:
        out["overall"] = {"delta_p": _summary(ev.pred_minus_true, ev.pred_minus_true_squared),
                          "delta_p_no_weighting": _summary(ev.pred_minus_true_deltap_crude, ev.pred_minus_true_squared_deltap_crude),
                          "p": _summary(ev.pred_minus_true_p, ev.pred_minus_true_squared_p)}
    return out


def main_gradP(delta=5e-3, model_directory="model_1.h5", shape=128, avance=None, var_p=0.95, var_in=0.95, max_number_PC=512,
               hdf5_path="dataset_gradP_cil.hdf5", plot_intermediate_fields=True, save_plots=True, show_plots=False,
               apply_filter=False, sims=(0,), n_ts=5, device: int = 0, artifact_dir: str = None):
    """``main`` of the U_to_gradP evaluator (Eval_dual_Dense_onlycil.py:692-744) with its hard-coded inputs as defaults
    (``avance = int(0.75 * shape)``, simulations ``[0]``, five time steps): computeOnlyOnce + timeStep per frame, then
    BIAS / RMSE / STDE [%] "for the sim" from the accumulated per-frame values (:735-742) -- printed like the reference and
    returned.  Plots are not produced."""
    if avance is None:
        avance = int(0.75 * shape)
    ev = EvaluationGradP(delta, shape, avance, var_p, var_in, hdf5_path, model_directory, max_number_PC, device=device,
                         artifact_dir=artifact_dir)
    ev.pred_minus_true, ev.pred_minus_true_squared = [], []
    out = {"sims": [], "frames": []}
    for sim in sims:
        ev.computeOnlyOnce(sim)
        for time in range(n_ts):
            ev.timeStep(sim, time, plot_intermediate_fields, save_plots, show_plots, apply_filter)
            out["frames"].append(dict(sim=sim, time=time, **{k: dict(v) for k, v in ev.last_metrics.items()}))
        # like the reference: over everything accumulated so far, not only this simulation (:735-737)
        s = _summary(ev.pred_minus_true, ev.pred_minus_true_squared)
        print("Metrics for the whole simulation:")
        print("BIAS for the sim: " + str(s["BIAS"]))
        print("RMSE for the sim: " + str(s["RMSE"]))
        print("STDE for the sim: " + str(s["STDE"]))
        out["sims"].append(s)
    return out


class EvaluationPoisson(Evaluation):
    """``pressureSM_Poisson.SM_call.Evaluation`` (pressureSM_Poisson/SM_call.py:71-118): the deltas layout fed with
    four channels (arcsinh-smoothed Poisson source term, dUx, dUy, SDF; mask = channel 3).

    ``max_abs`` = (max_abs_Poisson_term_1, max_abs_delta_Ux, max_abs_delta_Uy, max_abs_dist, max_abs_delta_p),
    the constants the reference reads from its ``maxs`` file."""
    variant = "deltas"
    c_in_expected, sdf_ch_expected = 4, 3

    def __init__(self, delta, shape, overlap, var_p, var_in, dataset_path, model_path, max_num_PC,
                 standardization_method, k, phis_fn, model: SurrogateModel = None, device: int = 0,
                 max_abs=None, artifact_dir: str = None):
        super().__init__(delta, shape, overlap, var_p, var_in, dataset_path, model_path, max_num_PC,
                         standardization_method, model, device, artifact_dir)
        model = self.artifacts
        if model.c_in != 4 or model.sdf_ch != 3:
            raise ValueError("the Poisson surrogate takes 4 input channels with the SDF in channel 3")
        self.k, self.phis_fn = k, phis_fn
        if max_abs is None:                                     # the five values of the `maxs` file (SM_call.py:124)
            max_abs = self.maxs if self.maxs is not None else (1.0, 1.0, 1.0, 1.0, 1.0)
        (self.max_abs_Poisson_term_1, self.max_abs_delta_Ux, self.max_abs_delta_Uy, self.max_abs_dist,
         self.max_abs_delta_p) = [float(v) for v in max_abs]

    def timeStep(self, sim, time, plot_intermediate_fields=False, save_plots=False, show_plots=False, apply_filter=False,
                 phi=1.0):
        """pressureSM_Poisson/SM_call.py:519-848 without plots and error prints: frame (sim, time) of the dataset ->
        ``field_deltap`` [Ny,Nx] = previous delta-p image + weighted change (:843-848), or 0 for an irrelevant step."""
        from . import formats
        if getattr(self, "tables", None) is None:
            raise RuntimeError("computeOnlyOnce has not been called")
        data, _, _ = formats.read_dataset(self.dataset_path, sim, time)
        d = data[0, 0, :self.indice]
        Ux, Uy, p = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        delta_U, delta_p = d[:, 5:7], d[:, 7:8]
        delta_U_prev, delta_p_prev = d[:, 8:10], d[:, 10:11]
        deltaU_changed = np.abs(delta_U - delta_U_prev).sum(axis=-1)
        deltaU_changed = deltaU_changed / deltaU_changed.max()
        U_max_norm = np.max(np.sqrt(np.square(Ux) + np.square(Uy)))
        deltaU_max_norm = np.max(np.sqrt(np.square(delta_U[:, 0:1]) + np.square(delta_U[:, 1:2])))
        if (deltaU_max_norm / U_max_norm) < 1e-4 or deltaU_max_norm < 1e-6 or U_max_norm < 1e-6:      # :562-567
            return 0
        cols = np.concatenate([Ux, Uy, delta_U, delta_p, p, deltaU_changed[:, None], delta_p_prev], axis=1).astype(np.float64)
        g = self._mesh_to_grid(cols)                                                                 # :577-600, NaNs kept
        U = float(U_max_norm)
        self.U_max_norm = U
        field_deltap = self.timeStep_grid(g[..., 0], g[..., 1], g[..., 2], g[..., 3], self.sdfunct[..., 0], phi, U,
                                          deltaU_change_grid=g[..., 6], deltaP_prev_grid=g[..., 7], apply_filter=apply_filter,
                                          apply_deltaU_change_wgt=True)                               # :831
        # ---- the three error blocks of :962-1043: delta-p (weighted), delta-p without the weighting, and p
        delta_p_grid = np.nan_to_num(g[..., 4] / pow(U, 2.0), nan=0.0) / self.max_abs_delta_p         # :687-711 (channel 4)
        p_grid = np.nan_to_num(g[..., 5], nan=0.0)                                                    # :692-702 (channel 5)
        self.cfd_results = delta_p_grid * self.max_abs_delta_p * pow(U, 2.0)                          # :849
        sd = np.nan_to_num(self.sdfunct[..., 0], nan=0.0) / self.max_abs_dist
        self.no_flow_bool = sd == 0                                                                   # :850
        self.p_pred = (p_grid - self.cfd_results) + field_deltap                                      # :907-908
        self._record_errors(field_deltap, self.cfd_results, self.no_flow_bool, "", "Error in delta_p")
        self._record_errors(self.deltap_res, self.cfd_results, self.no_flow_bool, "_deltap_crude", "Error in delta_p - no weighting")
        self._record_errors(self.p_pred, p_grid, self.no_flow_bool, "_p", "Error in p")
        return field_deltap

    def build_features(self, ux_grid, uy_grid, delta_ux_grid, delta_uy_grid, sdfunct, phi, U_max_norm) -> np.ndarray:
        """SM_call.py:588-711 (after the interpolation to the grid): -> grid [Ny,Nx,4] float32."""
        ny, nx = np.shape(ux_grid)
        sur = self._surrogate(ny, nx)
        return sur.poisson_features(ux_grid, uy_grid, delta_ux_grid, delta_uy_grid, sdfunct, phi, U_max_norm, self.k,
                                    (self.max_abs_Poisson_term_1, self.max_abs_delta_Ux, self.max_abs_delta_Uy,
                                     self.max_abs_dist))

    def timeStep_grid(self, ux_grid, uy_grid, delta_ux_grid, delta_uy_grid, sdfunct, phi, U_max_norm,
                      deltaU_change_grid=None, deltaP_prev_grid=None, apply_filter=False,
                      apply_deltaU_change_wgt=True) -> np.ndarray:
        """Grid-native body of ``timeStep`` (SM_call.py:588-848): features, surrogate, assembly, post-steps
        -> field_deltap [Ny,Nx] (``deltaP_prev_grid + change_in_deltap`` with the weighting, :843-848)."""
        grid = self.build_features(ux_grid, uy_grid, delta_ux_grid, delta_uy_grid, sdfunct, phi, U_max_norm)
        sur = self._surrogate(grid.shape[0], grid.shape[1])
        res = sur.solve(grid, out_scale=[self.max_abs_delta_p * U_max_norm ** 2])[0, :, :, 0]     # :816
        wgt = apply_deltaU_change_wgt and deltaU_change_grid is not None and deltaP_prev_grid is not None
        res, change = self._post_steps(sur, res, apply_filter, deltaU_change_grid, deltaP_prev_grid, wgt)
        self.deltap_res = res                                    # the assembled delta-p before the weighting (:833)
        return np.asarray(deltaP_prev_grid, np.float32) + change if wgt else res


class EvaluationGradP(Evaluation):
    """``U_to_gradP`` ``Evaluation`` (Eval_dual_Dense_onlycil.py:30-66)."""
    variant = "gradp"

    def __init__(self, delta, shape, avance, var_p, var_in, hdf5_path, model_path, max_number_PC,
                 model: SurrogateModel = None, device: int = 0, artifact_dir: str = None):
        super().__init__(delta, shape, avance, var_p, var_in, hdf5_path, model_path, max_number_PC,
                         model.scaler_kind if model is not None else "max_abs", model, device, artifact_dir)
        self.avance = avance
        self.hdf5_path = hdf5_path

    def computeOnlyOnce(self, sim):
        """Eval_dual_Dense_onlycil.py:160-253: 2-digit bounds, box = the ``top`` patch, every 2nd boundary point,
        column 5 decides interpolability.  Returns 0."""
        from . import formats
        from .geometry import build_geometry_evaluator
        data, top_b, obst_b = formats.read_dataset(self.hdf5_path, sim, 0)
        self.indice = formats.first_index(data[0, 0, :, 0], formats.PAD_VALUE)
        cells = np.asarray(data[0, 0, :self.indice], np.float64)
        top32 = top_b[0, 0, :formats.first_index(top_b[0, 0, :, 0], formats.PAD_VALUE)]
        obst32 = obst_b[0, 0, :formats.first_index(obst_b[0, 0, :, 0], formats.PAD_VALUE)]
        t = build_geometry_evaluator(cells[:, 3:5], cells[:, 5], np.asarray(top32, np.float64), np.asarray(obst32, np.float64),
                                     self.delta, every=2, round_digits=2, box="top")
        # float32 like `np.max(top[:,0])` of the float32 file data (:192): used in float32 arithmetic by timeStep
        self.max_x, self.max_y = np.max(top32[:, 0]), np.max(top32[:, 1])
        self.min_x, self.min_y = np.min(top32[:, 0]), np.min(top32[:, 1])
        self.grid_shape_y, self.grid_shape_x = t.ny, t.nx
        self.vert, self.weights, self.indices, self.sdfunct = t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct[:, :, None]
        self.X0_min = t.x0
        sur = self._surrogate(t.ny, t.nx)
        v1, w1 = np.ascontiguousarray(t.vtx_m2g, np.int32), _f64(t.wts_m2g)
        idx, sdf = np.ascontiguousarray(t.indices, np.int32), _f64(t.sdfunct)
        mx = _f64(np.asarray(self.maxs, np.float64)[:4])
        sur._chk(sur.lib.psm_set_geometry(sur.h, int(self.indice), t.ny, t.nx, _p(v1, C.c_int32), _p(w1, C.c_double),
                                          _p(idx, C.c_int32), _p(sdf, C.c_double), None, None, _p(mx, C.c_double), 1, 1, 0.05))
        self.tables = t
        self._bind_simulation_geometry(sur, t.sdfunct, float(mx[2]))
        return 0

    def timeStep(self, sim, time, plot_intermediate_fields=False, save_plots=False, show_plots=False, apply_filter=False):
        """Eval_dual_Dense_onlycil.py:418-640 without plots and error prints: frame (sim, time) -> the integrated
        pressure image [Ny,Nx] (``field``).  Kept: ``self.grid`` (6 channels), ``self.gradP`` [Ny,Nx,2]."""
        from . import formats
        if getattr(self, "tables", None) is None:
            raise RuntimeError("computeOnlyOnce has not been called")
        data, _, _ = formats.read_dataset(self.hdf5_path, sim, time)
        d = data[0, 0, :self.indice]                                  # float32, normalised in float32 like the reference
        Ux, Uy, p, dPdx, dPdy = d[:, 0:1], d[:, 1:2], d[:, 2:3], d[:, 6:7], d[:, 7:8]
        U_max_norm = np.max(np.sqrt(np.square(Ux) + np.square(Uy)))                       # :438
        cols = np.concatenate([Ux / U_max_norm, Uy / U_max_norm,                        # :443-444
                               dPdx * (self.max_x - self.min_x) / pow(U_max_norm, 2.0),   # :440
                               dPdy * (self.max_y - self.min_y) / pow(U_max_norm, 2.0),   # :441
                               p / pow(U_max_norm, 2.0)], axis=1).astype(np.float64)      # :442
        g = self._mesh_to_grid(cols)
        mx = [float(v) for v in self.maxs[:5]]
        grid = np.zeros((self.grid_shape_y, self.grid_shape_x, 6))
        grid[..., 0:2] = g[..., 0:2]
        grid[..., 2] = self.sdfunct[..., 0]
        grid[..., 3:6] = g[..., 2:5]
        grid[np.isnan(grid)] = 0                                                          # :461
        for ch in range(5):
            grid[..., ch] /= mx[ch]                                                       # :463-467
        self.grid = grid
        self.U_max_norm = float(U_max_norm)
        sur = self._surrogate(*grid.shape[:2])
        gradP = sur.solve(grid[..., :3])[0]                                               # :470-544
        if apply_filter:                                                                 # :366-367, per field
            gradP = np.stack([sur.gaussian_filter(gradP[..., c], (10, 10)) for c in range(2)], axis=-1)
        self.gradP = gradP
        xl = np.linspace(self.min_x, self.max_x, grid.shape[1])                            # :591-592
        yl = np.linspace(self.min_y, self.max_y, grid.shape[0])
        solid = self.sdfunct[200, :, 0] == 0                                              # :594 (row 200 is hard-wired)
        center_p_x = int(((xl[solid].max() + xl[solid].min()) / 2 - self.X0_min) / self.delta)
        center_p_y = 200
        sur.set_integration(self.sdfunct[..., 0], center_p_y, center_p_x, float(np.diff(xl)[0]), float(np.diff(yl)[0]))
        self.center_p_x, self.center_p_y = center_p_x, center_p_y
        field = sur.integrate_gradp(gradP)                                                # :597-628
        # ---- error block of :667-687.  The reference takes its "true" values from grid channel 3 -- the normalised dP/dx
        # label, not the pressure of channel 5 that its plot shows (:644-650) -- and that is what it accumulates in
        # pred_minus_true / pred_minus_true_squared; reproduced as written, with the comparison against the pressure label
        # kept beside it (last_metrics['p']).
        no_flow = grid[..., 2] == 0
        self.no_flow_bool = no_flow
        self._record_errors(field, grid[..., 3], no_flow, "", "reference metric (grid channel 3)")
        self.last_metrics["reference"] = self.last_metrics.pop("delta_p")
        self.last_metrics["p"] = error_metrics(field, grid[..., 5], no_flow)
        return field

    def timeStep_grid(self, grid: np.ndarray) -> np.ndarray:
        """Eval_dual_Dense_onlycil.py:470-547: -> [Ny,Nx,2] = (res_dPdx, res_dPdy)."""
        return self._surrogate(grid.shape[0], grid.shape[1]).solve(grid)[0]

    def assemble_prediction(self, field, array, indices_list, n_x, n_y, apply_filter, shape_x, shape_y):
        """Eval_dual_Dense_onlycil.py:255: one channel ('dp_dx' | 'dp_dy') of decoded blocks."""
        if field not in ("dp_dx", "dp_dy"):
            raise ValueError(field)
        blocks, nx_, ny_ = layout(self.variant, shape_y, shape_x, self.shape, self.avance)
        if (nx_, ny_) != (n_x, n_y) or len(indices_list) != len(blocks):
            raise ValueError("n_x / n_y / indices_list do not match this grid")
        sur = self._surrogate(shape_y, shape_x)
        grid = _grid_from_blocks(np.asarray(self.x_array, np.float32), blocks, shape_y, shape_x)
        a = np.asarray(array, np.float32)
        both = np.zeros(a.shape + (2,), np.float32)
        ch = 0 if field == "dp_dx" else 1
        both[..., ch] = a
        res = sur.reassemble(grid, both)[..., ch]
        if apply_filter:                                          # Eval_dual_Dense_onlycil.py:366-367 (returns the 2-D array)
            return sur.gaussian_filter(res, (10, 10))
        return res[None, :, :, None]


class SolverModule:
    """Chapter-5 ``python_module`` (python_module.py): ``init_func`` / ``py_func`` with the
    reference's signatures on one rank (the serial solver, singleCore/test_Case/python_module.py:
    139,199; in the parallel solver the mpi4py gather/scatter of python_module.py:179-185,258,511
    stays around these calls).  ``maxs`` = (max_abs_Ux, max_abs_Uy, max_abs_dist, max_abs_p) of
    the ``maxs`` file (python_module.py:106-109)."""

    def __init__(self, model: SurrogateModel, maxs=(1.0, 1.0, 1.0, 1.0), device: int = 0, delta: float = 5e-3,
                 geometry: str = "scipy"):
        """``geometry``: who builds init_func's tables -- 'scipy' (default in this Python mirror: the routines the reference
        itself calls, qhull Delaunay in both directions, so that the tables -- including qhull's arbitrary choice of diagonal
        in every cocircular lattice square of the grid -> mesh step -- are the reference's) or 'native' (the library's C++
        builder behind ``psm_init_geometry``, csrc/psm_geometry.cpp: what a C++ solver gets; no SciPy).  The two differ only
        where the reference's own result is an accident of qhull (include/psm.h): fixed lattice diagonals -- a second-order
        difference on smooth fields, O(1) per cell on the white-noise fields of randomly initialised test networks
        (tests/measure/mesh_native_vs_scipy.py) -- and the simplex used for lattice points outside the hull."""
        if geometry not in ("scipy", "native"):
            raise ValueError("geometry must be 'scipy' or 'native'")
        self.model, self.maxs, self.device, self.delta = model, tuple(float(v) for v in maxs), device, delta
        self.geometry = geometry
        self._sur = None
        self.tables = None

    # -- grid-native body (python_module.py:299-473)
    def py_func_grid(self, grid: np.ndarray) -> np.ndarray:
        if self._sur is None or (self._sur.ny, self._sur.nx) != grid.shape[:2]:
            if self._sur is not None:
                self._sur.close()
            self._sur = GridSurrogate(self.model, grid.shape[0], grid.shape[1], 1, self.device)
        return self._sur.solve(grid)[0, :, :, 0]

    # -- the solver boundary
    def init_func(self, array, top_boundary, obst_boundary, placeholder=0):
        """python_module.py:172: one-time tables (host, SciPy qhull like the reference), handed
        to the GPU library with psm_set_geometry.  Returns 0."""
        from .geometry import build_geometry
        if self.geometry == "native":
            return self._init_func_native(array, top_boundary, obst_boundary)
        t = build_geometry(np.asarray(array, np.float64), top_boundary, obst_boundary, self.delta)
        if self._sur is not None:
            self._sur.close()
        self._sur = GridSurrogate(self.model, t.ny, t.nx, 1, self.device)
        v1, w1 = np.ascontiguousarray(t.vtx_m2g, np.int32), _f64(t.wts_m2g)
        v2, w2 = np.ascontiguousarray(t.vtx_g2m, np.int32), _f64(t.wts_g2m)
        idx, sdf, mx = np.ascontiguousarray(t.indices, np.int32), _f64(t.sdfunct), _f64(self.maxs)
        self._sur._chk(self._sur.lib.psm_set_geometry(
            self._sur.h, int(np.asarray(array).shape[0]), t.ny, t.nx, _p(v1, C.c_int32), _p(w1, C.c_double),
            _p(idx, C.c_int32), _p(sdf, C.c_double), _p(v2, C.c_int32), _p(w2, C.c_double), _p(mx, C.c_double),
            0, 0, 0.05))
        self.tables = t
        return 0

    def _init_func_native(self, array, top_boundary, obst_boundary):
        """psm_set_case + psm_init_geometry: the tables are built inside the library (csrc/psm_geometry.cpp)."""
        a, t, o = _f64(array), _f64(top_boundary), _f64(obst_boundary)
        lib = _lib.load()
        ny, nx = C.c_int32(), C.c_int32()
        if lib.psm_geometry_shape(_p(a, C.c_double), a.shape[0], self.delta, C.byref(ny), C.byref(nx), None):
            raise ValueError(lib.psm_geometry_last_error().decode())
        if self._sur is not None:
            self._sur.close()
        self._sur = GridSurrogate(self.model, ny.value, nx.value, 1, self.device)
        mx = _f64(self.maxs)
        self._sur._chk(lib.psm_set_case(self._sur.h, _p(mx, C.c_double), self.delta, 10, 0.05))
        self._sur._chk(lib.psm_init_geometry(self._sur.h, _p(a, C.c_double), a.shape[0], _p(t, C.c_double), t.shape[0],
                                             _p(o, C.c_double), o.shape[0], 0))
        self.tables = "native"
        return 0

    def pin(self, array: np.ndarray, out: np.ndarray = None):
        """psm_pin_buffers: register the caller's persistent [N,5] float64 array (and optionally a persistent output
        array) for direct DMA; later ``py_func(array, out=out)`` calls with exactly these arrays skip the staging copies.
        The caller keeps both arrays alive until ``unpin()`` / a new ``init_func``."""
        if array.dtype != np.float64 or not array.flags.c_contiguous or (out is not None and (out.dtype != np.float64 or not out.flags.c_contiguous)):
            raise ValueError("contiguous float64 arrays expected")
        self._sur._chk(self._sur.lib.psm_pin_buffers(self._sur.h, _p(array, C.c_double), _p(out, C.c_double) if out is not None else None))
        self._pinned = (array, out)

    def unpin(self):
        self._sur._chk(self._sur.lib.psm_unpin_buffers(self._sur.h))
        self._pinned = None

    def py_func_begin(self, array_in, out: np.ndarray = None):
        """First half of :meth:`py_func` (psm_solve_begin): enqueue the step and return; several SolverModules (cases)
        can be advanced from one thread by calling begin on all of them, then :meth:`py_func_end` on each."""
        if self.tables is None:
            raise RuntimeError("init_func has not been called")
        a = _f64(array_in)
        if a.ndim != 2 or a.shape[1] != 5:
            raise ValueError("array must be [N,5] = (Ux, Uy, Cx, Cy, p)")
        if out is None:
            out = np.empty(a.shape[0], np.float64)
        self._sur._chk(self._sur.lib.psm_solve_begin(self._sur.h, _p(a, C.c_double), a.shape[0], 0, _p(out, C.c_double)))
        self._inflight = (a, out)                       # keeps both arrays alive until the end call (set only once the step is in flight)

    def py_func_end(self) -> np.ndarray:
        self._sur._chk(self._sur.lib.psm_solve_end(self._sur.h))
        a, out = self._inflight
        self._inflight = None
        return out

    def py_func(self, array_in, placeholder=0, out: np.ndarray = None) -> np.ndarray:
        """python_module.py:249: cells [N,5] float64 -> p [N] float64."""
        if self.tables is None:
            raise RuntimeError("init_func has not been called")
        a = _f64(array_in)
        if a.ndim != 2 or a.shape[1] != 5:
            raise ValueError("array must be [N,5] = (Ux, Uy, Cx, Cy, p)")
        if out is None:
            out = np.empty(a.shape[0], np.float64)
        self._sur._chk(self._sur.lib.psm_solve(self._sur.h, _p(a, C.c_double), a.shape[0], int(placeholder),
                                                _p(out, C.c_double)))
        return out
