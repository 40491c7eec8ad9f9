"""CPU oracle of the convolutional surrogate path (SURVEY.md §8 row a-conv) -- TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  The project's north star names a Conv2D / U-Net forward pass, but nothing in the
reference repository defines one: its CNN folders are empty placeholders
(Thesis_Work/Chapter4/README.md:3) and the model called "U-Net" at python_module.py:131 contains only Dense
layers.  There are no weights, no layer list and no outputs to compare with.  This file therefore states a
build-defined network ("UNet-S", the spec SURVEY.md §8 gives) in plain NumPy; tests/test_unet.py cross-checks
it against ``torch.nn.functional.conv2d`` / ``max_pool2d`` / ``interpolate`` on the CPU and then uses it as
the checker of the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.

UNet-S (NHWC, float32; Keras conventions: kernels HWIO ``[kh, kw, c_in, c_out]``, 'same' zero padding):

  enc_l  (l = 0..L-1): [2x2 max-pool of enc_{l-1} if l > 0] -> conv3x3(w_l)+ReLU -> conv3x3(w_l)+ReLU
  dec_l  (l = L-2..0): concat(nearest-neighbour 2x upsample of the level below, enc_l) on the channel axis
                       (upsampled channels first) -> conv3x3(w_l)+ReLU -> conv3x3(w_l)+ReLU
  head               : conv1x1(w_0 -> c_out), linear

with widths w = (16, 32, 64, 128, 256): 18 3x3 convolutions + the head, 7.0 GFLOP for a 256x256x3 image.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

WIDTHS_S = (16, 32, 64, 128, 256)


@dataclass
class ConvSpec:
    name: str
    k: int          # kernel edge (3 or 1)
    c_in: int
    c_out: int
    level: int      # resolution level of the convolution's output (0 = full resolution)
    src: str        # 'input' | 'prev' | 'pool' (2x2 max-pool of prev) | 'up+skip' (upsample(prev) ++ enc_level)
    relu: bool


def unet_specs(c_in: int = 3, widths=WIDTHS_S, c_out: int = 1) -> List[ConvSpec]:
    L = len(widths)
    s: List[ConvSpec] = []
    for l in range(L):
        cin = c_in if l == 0 else widths[l - 1]
        s.append(ConvSpec(f"enc{l}a", 3, cin, widths[l], l, "input" if l == 0 else "pool", True))
        s.append(ConvSpec(f"enc{l}b", 3, widths[l], widths[l], l, "prev", True))
    for l in range(L - 2, -1, -1):
        s.append(ConvSpec(f"dec{l}a", 3, widths[l + 1] + widths[l], widths[l], l, "up+skip", True))
        s.append(ConvSpec(f"dec{l}b", 3, widths[l], widths[l], l, "prev", True))
    s.append(ConvSpec("head", 1, widths[0], c_out, 0, "prev", False))
    return s


def he_weights(specs: List[ConvSpec], seed: int = 7) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Seeded He-normal kernels (HWIO) and small biases, float32."""
    rng = np.random.default_rng(seed)
    out = []
    for sp in specs:
        fan_in = sp.k * sp.k * sp.c_in
        W = (rng.standard_normal((sp.k, sp.k, sp.c_in, sp.c_out)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        b = (rng.standard_normal(sp.c_out) * 0.05).astype(np.float32)
        out.append((W, b))
    return out


def conv2d_same(x: np.ndarray, W: np.ndarray, b: np.ndarray, relu: bool) -> np.ndarray:
    """x [H,W,Cin] -> [H,W,Cout]; zero 'same' padding; accumulation in float64, result float32."""
    k = W.shape[0]
    r = k // 2
    H, Wd, _ = x.shape
    xp = np.zeros((H + 2 * r, Wd + 2 * r, x.shape[2]), np.float64)
    xp[r:r + H, r:r + Wd] = x
    acc = np.zeros((H, Wd, W.shape[3]), np.float64)
    for ky in range(k):
        for kx in range(k):
            acc += xp[ky:ky + H, kx:kx + Wd] @ W[ky, kx].astype(np.float64)
    acc += b.astype(np.float64)
    if relu:
        acc = np.maximum(acc, 0.0)
    return acc.astype(np.float32)


def max_pool2(x: np.ndarray) -> np.ndarray:
    H, W, C = x.shape
    return x[:H - H % 2, :W - W % 2].reshape(H // 2, 2, W // 2, 2, C).max(axis=(1, 3))


def upsample2(x: np.ndarray) -> np.ndarray:
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


def bf16_round(x) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def unet_forward(grid: np.ndarray, weights, widths=WIDTHS_S, return_all: bool = False, precision: str = "f32"):
    """grid [H,W,c_in] float32 (H, W multiples of 2**(L-1)) -> [H,W,c_out] float32; activations are rounded to
    float32 after every layer like the device path stores them.  precision 'bf16': the input of every 3x3
    convolution (after pooling / upsampling / concatenation) and its kernel are rounded to bf16 first, products
    exact, accumulation wide -- the device's bf16 operand path; biases, the 1x1 head and the stored activations
    stay float32."""
    L = len(widths)
    rnd = bf16_round if precision == "bf16" else (lambda v: v)
    conv3 = lambda x, Wb: conv2d_same(rnd(x), rnd(Wb[0]), Wb[1], True)
    x = np.asarray(grid, np.float32)
    acts, enc, i = [], [], 0
    for l in range(L):
        if l > 0:
            x = max_pool2(x)
        for _ in range(2):
            x = conv3(x, weights[i]); acts.append(x); i += 1
        enc.append(x)
    for l in range(L - 2, -1, -1):
        x = np.concatenate([upsample2(x), enc[l]], axis=-1)
        for _ in range(2):
            x = conv3(x, weights[i]); acts.append(x); i += 1
    x = conv2d_same(x, *weights[i], False); acts.append(x)
    return (x, acts) if return_all else x


def unet_flops(H: int, W: int, c_in: int = 3, widths=WIDTHS_S, c_out: int = 1) -> int:
    tot = 0
    for sp in unet_specs(c_in, widths, c_out):
        tot += 2 * (H >> sp.level) * (W >> sp.level) * sp.k * sp.k * sp.c_in * sp.c_out
    return tot
