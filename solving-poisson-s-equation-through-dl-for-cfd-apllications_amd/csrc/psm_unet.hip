// psm_unet.hip -- direct 3x3 convolution for the U-Net surrogate path (north star: "im2col-free direct 2-D
// convolution with LDS-staged input tiles and coalesced HBM loads, MFMA for the Conv2D channel contractions,
// fused bias+ReLU, nearest-neighbour upsample"; SURVEY.md §8 row a-conv, build plan item 7).  The reference has
// no convolutional surrogate: the network is the build-defined UNet-S of oracle/unet_oracle.py.
//
// One kernel serves every 3x3 layer.  A workgroup owns an output tile of TH rows x 16 columns and NCT channel
// tiles of 16; it walks the input channels in chunks of 16.  Per chunk it stages
//   * the input tile with its one-pixel halo, (TH+2) x 18 pixels x 16 channels, into LDS -- read through the
//     layer's source transform, so neither the 2x2 max-pool, nor the 2x nearest-neighbour upsample, nor the
//     skip concatenation is ever materialised in HBM (zero 'same' padding = out-of-image pixels read as 0),
//   * the chunk's weights, 9 taps x NCT x 1 KiB, already in MFMA operand order;
// then every wave runs v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation) over the nine taps:
//   A: lane l holds pixel (l & 15) of its row, channels 4*(l >> 4) + j  (one ds_read_b128 feeds 4 MFMAs)
//   B: lane l holds output channel (l & 15), the same four input channels (one ds_read_b128)
//   D: lane l, register r holds pixel 4*(l >> 4) + r, output channel (l & 15)
// Epilogue: bias, ReLU, NHWC store (16 consecutive channels = 64 B per pixel).
#include "psm_unet.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int TW = 16;            // tile width (pixels) = MFMA rows
constexpr int CC = 16;            // input channels per chunk
constexpr int LDC = CC + 4;       // LDS pixel stride (floats): 16-B slots rotate from pixel to pixel

// one finished value group: sum of the producer's partial-sum slabs (+ its bias, ReLU) or the plain activation
template <bool ALIGNED4>
__device__ __forceinline__ f32x4 read4(const float* p, int ks, int64_t slab, const float* pbias, int ch, int nvalid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ALIGNED4) {
    v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < ks; ++s) v += *reinterpret_cast<const f32x4*>(p + s * slab);
    if (ks > 1) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(pbias + ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] + b[j], 0.f);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < nvalid) {
        float t = p[j];
        for (int s = 1; s < ks; ++s) t += p[s * slab + j];
        v[j] = ks > 1 ? fmaxf(t + pbias[ch + j], 0.f) : t;
      }
    }
  }
  return v;
}

// four consecutive channels [ch, ch+4) of the concatenated input at output-resolution pixel (y, x)
template <bool ALIGNED4>
__device__ __forceinline__ f32x4 fetch4(const PsmConvArgs& a, const float* in0, const float* in1, int y, int x, int ch) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (y < 0 || y >= a.H || x < 0 || x >= a.W) return v;                       // zero padding
  if (ch < a.c0) {
    const int nv = min(4, a.c0 - ch);
    if (a.mode0 == PSM_SRC_SAME) {
      v = read4<ALIGNED4>(in0 + ((int64_t)y * a.W0 + x) * a.c0 + ch, a.ks0, a.slab0, a.pbias0, ch, nv);
    } else if (a.mode0 == PSM_SRC_UPSAMPLE) {
      v = read4<ALIGNED4>(in0 + ((int64_t)(y >> 1) * a.W0 + (x >> 1)) * a.c0 + ch, a.ks0, a.slab0, a.pbias0, ch, nv);
    } else {
      const float* p = in0 + ((int64_t)(2 * y) * a.W0 + 2 * x) * a.c0 + ch;
      const f32x4 q0 = read4<ALIGNED4>(p, a.ks0, a.slab0, a.pbias0, ch, nv);
      const f32x4 q1 = read4<ALIGNED4>(p + a.c0, a.ks0, a.slab0, a.pbias0, ch, nv);
      const f32x4 q2 = read4<ALIGNED4>(p + (int64_t)a.W0 * a.c0, a.ks0, a.slab0, a.pbias0, ch, nv);
      const f32x4 q3 = read4<ALIGNED4>(p + (int64_t)a.W0 * a.c0 + a.c0, a.ks0, a.slab0, a.pbias0, ch, nv);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaxf(q0[j], q1[j]), fmaxf(q2[j], q3[j]));
    }
    if (!ALIGNED4 && nv < 4 && a.c1 > 0) {           // a group straddling the concatenation seam (unaligned widths only)
      for (int j = nv; j < 4; ++j)
        if (ch + j - a.c0 < a.c1) {
          const f32x4 t = read4<false>(in1 + ((int64_t)y * a.W + x) * a.c1 + (ch + j - a.c0), a.ks1, a.slab1, a.pbias1, ch + j - a.c0, 1);
          v[j] = t[0];
        }
    }
  } else if (ch - a.c0 < a.c1) {
    const int c = ch - a.c0;
    v = read4<ALIGNED4>(in1 + ((int64_t)y * a.W + x) * a.c1 + c, a.ks1, a.slab1, a.pbias1, c, min(4, a.c1 - c));
  }
  return v;
}

// TH: tile rows; WM: rows per wave; NCT: channel tiles per workgroup; WN: channel tiles per wave
template <int TH, int WM, int NCT, int WN, bool ALIGNED4>
__global__ __launch_bounds__(256) void psm_conv3x3_kernel(PsmConvArgs a, int co_groups) {
  __shared__ __attribute__((aligned(16))) float in_tile[(TH + 2) * (TW + 2) * LDC];
  __shared__ __attribute__((aligned(16))) f32x4 w_tile[9 * NCT * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zz = blockIdx.z / a.ksplit, split = blockIdx.z - zz * a.ksplit;
  const int cs = zz / co_groups, cog = zz - cs * co_groups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* in0 = a.in0 + (int64_t)cs * a.in0_case;
  const float* in1 = a.in1 ? a.in1 + (int64_t)cs * a.in1_case : nullptr;
  // wave -> (rows, channel tiles) of the workgroup tile
  const int row_w = (TH == 4 * WM) ? wave * WM : 0;           // pixel-major: waves stacked along the rows
  const int ct_w = (TH == 4 * WM) ? 0 : wave * WN;            // channel-major: waves along the channel tiles
  const int px = lane & 15, kq = lane >> 4;
  f32x4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* wsrc = a.wpack + (int64_t)cog * a.n_chunks * (9 * NCT * 64);
  const int cps = (a.n_chunks + a.ksplit - 1) / a.ksplit;          // chunks per split
  const int g_end = min(a.n_chunks, (split + 1) * cps);
  for (int g = split * cps; g < g_end; ++g) {
    // ---- stage the chunk: weights (contiguous 9*NCT KiB) and the input tile with its halo
    for (int q = tid; q < 9 * NCT * 64; q += 256) {
      const float4 w = wsrc[(int64_t)g * (9 * NCT * 64) + q];
      w_tile[q] = (f32x4){w.x, w.y, w.z, w.w};
    }
    for (int q = tid; q < (TH + 2) * (TW + 2) * (CC / 4); q += 256) {
      const int pos = q >> 2, c4 = q & 3;
      const int r = pos / (TW + 2), c = pos - r * (TW + 2);
      const f32x4 v = fetch4<ALIGNED4>(a, in0, in1, y0 - 1 + r, x0 - 1 + c, g * CC + 4 * c4);
      *reinterpret_cast<f32x4*>(&in_tile[pos * LDC + 4 * c4]) = v;
    }
    __syncthreads();
    // ---- nine taps x (WM rows) x (WN channel tiles) x 4 MFMAs
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      f32x4 av[WM], bv[WN];
#pragma unroll
      for (int m = 0; m < WM; ++m)
        av[m] = *reinterpret_cast<const f32x4*>(&in_tile[((row_w + m + ky) * (TW + 2) + px + kx) * LDC + 4 * kq]);
#pragma unroll
      for (int n = 0; n < WN; ++n) bv[n] = w_tile[(tap * NCT + ct_w + n) * 64 + lane];
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          acc[m][n] = MFMA16(av[m][0], bv[n][0], acc[m][n]);
          acc[m][n] = MFMA16(av[m][1], bv[n][1], acc[m][n]);
          acc[m][n] = MFMA16(av[m][2], bv[n][2], acc[m][n]);
          acc[m][n] = MFMA16(av[m][3], bv[n][3], acc[m][n]);
        }
    }
    __syncthreads();
  }
  // ---- epilogue: bias + ReLU (split-K: the raw partial sum into this split's slab), NHWC store
  float* out = a.out + (int64_t)cs * a.out_case + (int64_t)split * a.out_slab;
  const bool fin = a.ksplit == 1;
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int co = (cog * NCT + ct_w + n) * 16 + (lane & 15);
    const float b = (fin && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int y = y0 + row_w + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = x0 + 4 * kq + r;
        float v = acc[m][n][r] + b;
        if (fin && a.relu) v = fmaxf(v, 0.f);
        if (y < a.H && x < a.W && co < a.cout) out[((int64_t)y * a.W + x) * a.cout + co] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void psm_head1x1_kernel(PsmHeadArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_pix) return;
  const float* x = a.in + p * a.c_in;
  for (int co = 0; co < a.c_out; ++co) {
    float acc = 0.f;
    for (int c = 0; c < a.c_in; ++c) acc = fmaf(x[c], a.w[c * a.c_out + co], acc);
    a.out[p * a.c_out + co] = acc + a.bias[co];
  }
}

}  // namespace

hipError_t psm_launch_conv3x3(const PsmConvArgs& a, int arrangement, int nct, int n_cases, hipStream_t st) {
  const int cout_tiles = (a.cout + 15) / 16;
  const bool al = (a.c0 % 4 == 0) && (a.c1 % 4 == 0);
  if (arrangement == 0) {          // pixel-major: 8 rows x 16 columns, every wave 2 rows x all NCT channel tiles
    const int groups = (cout_tiles + nct - 1) / nct;
    const dim3 grid((a.W + TW - 1) / TW, (a.H + 7) / 8, n_cases * groups * a.ksplit);
#define PIX(N)                                                                                              \
    do {                                                                                                    \
      if (al) hipLaunchKernelGGL((psm_conv3x3_kernel<8, 2, N, N, true>), grid, dim3(256), 0, st, a, groups);  \
      else hipLaunchKernelGGL((psm_conv3x3_kernel<8, 2, N, N, false>), grid, dim3(256), 0, st, a, groups);    \
    } while (0)
    if (nct == 1) PIX(1); else if (nct == 2) PIX(2); else if (nct == 4) PIX(4); else return hipErrorInvalidValue;
#undef PIX
  } else {                         // channel-major: 2 rows x 16 columns, 4 waves = 4 channel tiles
    if (nct != 4) return hipErrorInvalidValue;
    const int groups = (cout_tiles + 3) / 4;
    const dim3 grid((a.W + TW - 1) / TW, (a.H + 1) / 2, n_cases * groups * a.ksplit);
    if (al) hipLaunchKernelGGL((psm_conv3x3_kernel<2, 2, 4, 1, true>), grid, dim3(256), 0, st, a, groups);
    else hipLaunchKernelGGL((psm_conv3x3_kernel<2, 2, 4, 1, false>), grid, dim3(256), 0, st, a, groups);
  }
  return hipGetLastError();
}

hipError_t psm_launch_head1x1(const PsmHeadArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_head1x1_kernel, dim3((unsigned)((a.n_pix + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}
