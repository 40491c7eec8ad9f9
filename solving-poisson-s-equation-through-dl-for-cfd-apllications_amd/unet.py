"""Host-side mirror of the convolutional surrogate path (include/psm_unet.h) -- the CNN / U-Net forward pass the
project's north star names.  The reference repository has no such network, so the class
follows the calling convention of the reference's Keras models (NHWC float32 images, Conv2D kernels
[kh, kw, c_in, c_out], 'same' padding) and the build-defined UNet-S layer list of oracle/unet_oracle.py.
Everything numeric runs in the HIP library; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib

WIDTHS_S = (16, 32, 64, 128, 256)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class UNetSurrogate:
    """weights: [(kernel[k,k,c_in,c_out] f32, bias[c_out] f32)] in the order enc0a, enc0b, ..., dec0a, dec0b, head."""

    def __init__(self, weights, ny: int, nx: int, c_in: int = 3, c_out: int = 1, widths: Sequence[int] = WIDTHS_S,
                 max_cases: int = 1, device: int = 0, precision: str = "f32", keep_activations: bool = False,
                 autotune: bool = False, choices=None):
        self.lib = _lib.load()
        self.ny, self.nx, self.c_in, self.c_out, self.max_cases = int(ny), int(nx), int(c_in), int(c_out), int(max_cases)
        w = np.ascontiguousarray(widths, np.int32)
        h = C.c_void_p()
        rc = self.lib.psm_unet_create(c_in, c_out, len(w), _p(w, C.c_int32), device, C.byref(h))
        if rc:
            raise _lib.PsmError(rc, (self.lib.psm_unet_last_error(None) or b"").decode())
        self.h = h
        try:
            n = self.lib.psm_unet_num_convs(self.h)
            if len(weights) != n:
                raise ValueError(f"{n} convolutions expected, {len(weights)} given")
            self.shapes = []
            for i, (W, b) in enumerate(weights):
                k, ci, co = C.c_int32(), C.c_int32(), C.c_int32()
                self._chk(self.lib.psm_unet_conv_shape(self.h, i, C.byref(k), C.byref(ci), C.byref(co)))
                if tuple(np.shape(W)) != (k.value, k.value, ci.value, co.value) or tuple(np.shape(b)) != (co.value,):
                    raise ValueError(f"convolution {i}: kernel {np.shape(W)} / bias {np.shape(b)}, expected "
                                     f"{(k.value, k.value, ci.value, co.value)} / {(co.value,)}")
                self.shapes.append((k.value, ci.value, co.value))
                self._chk(self.lib.psm_unet_set_conv(self.h, i, _p(_f32(W)), _p(_f32(b))))
            self._chk(self.lib.psm_unet_set_precision(self.h, _lib.PRECISIONS[precision]))
            # bf16 mode keeps the inner activation of a fused level pair on chip; True stores it too (activation(i) of every layer)
            self._chk(self.lib.psm_unet_keep_activations(self.h, 1 if keep_activations else 0))
            self._chk(self.lib.psm_unet_plan(self.h, self.ny, self.nx, self.max_cases))
            self.autotuned = None
            if autotune:                       # split-K depth per layer, measured on this GPU for max_cases cases (psm_unet_autotune)
                b, a = C.c_float(), C.c_float()
                self._chk(self.lib.psm_unet_autotune(self.h, self.max_cases, 20, C.byref(b), C.byref(a)))
                self.autotuned = {"us_before": float(b.value), "us_after": float(a.value),
                                  "ksplit": [int(self.lib.psm_unet_ksplit(self.h, i)) for i in range(n)],
                                  "plan": [self.plan_info(i) for i in range(n)]}
            if choices is not None:            # replay of a plan an earlier (autotuned) handle reported: bit-identical fields
                self.set_choices(choices)
        except Exception:
            self.close()
            raise

    @classmethod
    def from_keras_h5(cls, path: str, ny: int, nx: int, **kw) -> "UNetSurrogate":
        """Conv2D kernels / biases of a Keras HDF5 file (layers conv2d, conv2d_1, ... in creation order); channel
        counts and depth are inferred from the kernel shapes (formats.unet_layout_from_weights)."""
        from . import formats
        weights = formats.read_keras_conv_weights(path)
        c_in, widths, c_out = formats.unet_layout_from_weights(weights)
        return cls(weights, ny, nx, c_in=c_in, c_out=c_out, widths=widths, **kw)

    def _chk(self, rc):
        if rc:
            raise _lib.PsmError(rc, (self.lib.psm_unet_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.psm_unet_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, grid: np.ndarray) -> np.ndarray:
        """grid [Ny,Nx,c_in] or [n,Ny,Nx,c_in] -> field [n,Ny,Nx,c_out] float32."""
        g = np.asarray(grid)
        if g.ndim == 3:
            g = g[None]
        if g.ndim != 4 or g.shape[1:] != (self.ny, self.nx, self.c_in):
            raise ValueError(f"grid must be [n,{self.ny},{self.nx},{self.c_in}]")
        g = _f32(g)
        out = np.empty((g.shape[0], self.ny, self.nx, self.c_out), np.float32)
        self._chk(self.lib.psm_unet_forward(self.h, _p(g), g.shape[0], _p(out)))
        return out

    def forward_device(self, d_grid: int, n_cases: int, d_field: int, stream: int = 0):
        self._chk(self.lib.psm_unet_forward_device(self.h, d_grid, n_cases, d_field, stream))

    def synchronize(self):
        self._chk(self.lib.psm_unet_synchronize(self.h))

    def activation(self, idx: int, n_cases: int = 1) -> np.ndarray:
        """Output of convolution ``idx`` of the last forward pass -> [n, H_l, W_l, c_out]."""
        k, ci, co = self.shapes[idx]
        level = self._level(idx)
        shape = (n_cases, self.ny >> level, self.nx >> level, co)
        out = np.empty(shape, np.float32)
        self._chk(self.lib.psm_unet_read_activation(self.h, idx, _p(out), out.size))
        return out

    def _level(self, idx: int) -> int:
        L = (len(self.shapes) - 1 + 2) // 4 + 0          # 2L + 2(L-1) + 1 convolutions
        L = (len(self.shapes) + 1) // 4
        if idx < 2 * L:
            return idx // 2
        if idx == len(self.shapes) - 1:
            return 0
        return L - 2 - (idx - 2 * L) // 2

    def profile(self, d_grid: int, n_cases: int, d_field: int):
        """-> (ms per convolution, workgroups per convolution) of one instrumented forward pass."""
        ms = np.zeros(len(self.shapes), np.float32)
        wg = np.zeros(len(self.shapes), np.int32)
        self._chk(self.lib.psm_unet_profile(self.h, d_grid, n_cases, d_field, _p(ms), _p(wg, C.c_int32)))
        return ms, wg

    def get_choices(self):
        """The planner's per-layer choices (psm_unet_get_choices): [[split-K cap, tile, pair, x6]] per convolution -- what
        psm_unet_autotune decided; feed it to ``UNetSurrogate(..., choices=...)`` / ``set_choices`` to replay the plan."""
        n = len(self.shapes)
        buf = (C.c_int32 * (4 * n))()
        rc = self.lib.psm_unet_get_choices(self.h, buf, 4 * n)
        if rc != n:
            self._chk(rc if rc < 0 else -1)
        return [[int(buf[4 * i + j]) for j in range(4)] for i in range(n)]

    def set_choices(self, choices):
        flat = [int(v) for row in choices for v in row]
        buf = (C.c_int32 * len(flat))(*flat)
        self._chk(self.lib.psm_unet_set_choices(self.h, buf, len(flat)))

    def plan_info(self, idx: int):
        """(tile rows, channel tiles per workgroup, split-K, role) of convolution idx in the current plan; role & 3 = pair role
        (0 none, 1 leader, 2 computed by the leader's launch), role & 4 = x6 arithmetic (float32 mode), role & 8 = in-workgroup K split
        (eight-wave workgroups, bf16 mode)."""
        info = (C.c_int32 * 4)()
        self._chk(self.lib.psm_unet_plan_info(self.h, idx, info))
        return [int(v) for v in info]

    def time_kernels(self, d_grid: int, n_cases: int, d_field: int, steps: int = 20, quantiles: bool = False):
        """Dispatch-level timing -> per launch (first conv index, convs covered, kernel name, MEDIAN us) -- psm_unet_time_kernels_q;
        with ``quantiles`` a fifth element (p10 us, p90 us).  (The mean of psm_unet_time_kernels let one slow dispatch pick the
        "dominant" launch of a bench line.)"""
        n = len(self.shapes)
        us, p10, p90 = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        cnt = (C.c_int32 * n)()
        names = C.create_string_buffer(n * 64)
        self._chk(self.lib.psm_unet_time_kernels_q(self.h, d_grid, n_cases, d_field, steps, us, p10, p90, cnt, names))
        starts = [i for i in range(n) if cnt[i] > 0]
        out = []
        for j, i in enumerate(starts):
            end = starts[j + 1] if j + 1 < len(starts) else n
            rec = (i, list(range(i, end)), names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode(), float(us[i]))
            out.append(rec + ((float(p10[i]), float(p90[i])),) if quantiles else rec)
        return out

    def conv_flops(self, idx: int) -> int:
        k, ci, co = self.shapes[idx]
        lv = self._level(idx)
        return 2 * (self.ny >> lv) * (self.nx >> lv) * k * k * ci * co

    def conv_bytes(self, idx: int, act_bytes: int = 4, w_bytes: int = 4):
        """Algorithmic bytes of convolution idx for one case: (input activations, output activation, weights)."""
        k, ci, co = self.shapes[idx]
        lv = self._level(idx)
        L = (len(self.shapes) + 1) // 4
        hw = (self.ny >> lv) * (self.nx >> lv)
        last = idx == len(self.shapes) - 1
        if idx == 0:
            a_in = hw * ci * 4                                   # the float32 image
        elif idx < 2 * L and idx % 2 == 0:
            a_in = 4 * hw * ci * act_bytes                       # max-pool source at twice the resolution
        elif 2 * L <= idx < len(self.shapes) - 1 and (idx - 2 * L) % 2 == 0:
            up = self.shapes[idx - 1][2]
            a_in = (hw // 4) * up * act_bytes + hw * (ci - up) * act_bytes   # upsample source + skip
        else:
            a_in = hw * ci * act_bytes
        return a_in, hw * co * (4 if last else act_bytes), k * k * ci * co * (4 if last else w_bytes)

    @property
    def flops(self) -> int:
        return int(self.lib.psm_unet_flops(self.h))
