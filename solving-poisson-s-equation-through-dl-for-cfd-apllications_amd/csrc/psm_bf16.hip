// psm_bf16.hip -- bf16 operand path of the two PCA contractions (BASELINE config 4):
// bases and the (centred) block / coefficient operands are rounded to bf16 (round-to-nearest-
// even, v_cvt_pk_bf16_f32), products are exact and accumulated in f32 by
// v_mfma_f32_32x32x16_bf16.  Halves the streamed basis bytes and cuts the MFMA time 16x, so
// both kernels become pure HBM/MALL streams.  Same launch geometry, slab layout and epilogues
// as the f32 kernels (psm_kernels.hip); reference lines: PM:341-352 (encode), PM:365-366 /
// SMD:541-551 (decode).
//
// MFMA operand maps (wave64, 32x32x16): lane l (r = l&31, h = l>>5) holds A[r][8h+j] and
// B[8h+j][r], j = 0..7 (one 16-byte register group each); D as for the f32 32x32 form.
#include "psm_kernels.h"
#include "psm_devutil.h"

#include <hip/hip_ext.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256) void psm_encode_bf16_kernel(PsmEncodeArgs a) {
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per workgroup
  constexpr int G = KS / 16;                    // MFMA steps (16 k each)
  constexpr int LDA = KS + 8;                   // LDS row stride in bf16 (16-B slots rotate by an odd count per row)
  constexpr int Q = KS / 4;                     // 4-float pieces per activation row
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  const int r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
  const int64_t src_off = (int64_t)r * a.row_stride + (int64_t)c0 * C_IN;
  const int NT = a.NT;
  const int i = lane & 31, h = lane >> 5;
  const int ql = lane < Q ? lane : Q - 1;
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + (int64_t)s * KS + 4 * ql);
  const bf16x8* bpack = reinterpret_cast<const bf16x8*>(a.bpack);

  auto load_rows = [&](v4f (&x)[8], int m0, int row0) {
    int64_t rb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rb[u] = psm_row_base(a.row_base, min(m0 + row0 + wave + 4 * u, a.M - 1));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* src = a.grid + rb[u] + src_off + 4 * ql;
      if (ALIGNED) x[u] = *reinterpret_cast<const v4f*>(src);
      else x[u] = (v4f){src[0], src[1], src[2], src[3]};
    }
  };
  auto write_rows = [&](const v4f (&x)[8], int m0, int row0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = row0 + wave + 4 * u;
      const float keep = (m0 + row) < a.M ? 1.f : 0.f;
      bf16x4 v;
      v[0] = (__bf16)((x[u].x - mu.x) * keep); v[1] = (__bf16)((x[u].y - mu.y) * keep);
      v[2] = (__bf16)((x[u].z - mu.z) * keep); v[3] = (__bf16)((x[u].w - mu.w) * keep);
      if (lane < Q) *reinterpret_cast<bf16x4*>(&ldsb[row * LDA + 4 * lane]) = v;
    }
  };
  auto gemm_tile = [&](const bf16x8 (&b)[G], int mt, int t, int m0, bool store) {
    f32x16 acc = {0};
    const __bf16* arow = &ldsb[(mt * 32 + i) * LDA + 8 * h];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + 16 * g);
      acc = MFMA_BF16(av, b[g], acc);
    }
    if (store) {
      float* out = a.part + ((int64_t)s * a.Mpad + m0 + mt * 32) * a.ldp + t * 32 + i;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg];
    }
  };

  bf16x8 b[G];
  int cur_t = -1;
  for (int m0 = 0; m0 < a.Mpad; m0 += 32 * PSM_MT_CHUNK) {
    const int rows = min(32 * PSM_MT_CHUNK, a.Mpad - m0);
    v4f x0[8];
    load_rows(x0, m0, 0);                       // activation rows first, then the weight stream
    __builtin_amdgcn_sched_barrier(0);
    if (m0 == 0 && wave < NT) {
      const bf16x8* p = bpack + (((int64_t)s * NT + wave) * G) * 64 + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = p[g * 64];
      cur_t = wave;
    }
    __builtin_amdgcn_sched_barrier(0);
    write_rows(x0, m0, 0);
    for (int row0 = 32; row0 < rows; row0 += 32) {
      v4f x[8];
      load_rows(x, m0, row0);
      write_rows(x, m0, row0);
    }
    __syncthreads();
    for (int t = wave; t < NT; t += 4) {
      if (t != cur_t) {
        const bf16x8* p = bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) b[g] = p[g * 64];
        cur_t = t;
      }
      for (int mt = 0; mt < rows / 32; ++mt) gemm_tile(b, mt, t, m0, true);
    }
    __syncthreads();
  }
}

hipError_t psm_launch_encode_bf16(const PsmEncodeArgs& a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  const int rows = a.Mpad < 32 * PSM_MT_CHUNK ? a.Mpad : 32 * PSM_MT_CHUNK;
  const size_t lds = (size_t)rows * (PSM_PIX_PER_SLICE * a.c_in + 8) * 2;
#define ENC2(C, AL)                                                                                          \
  if (ev_start) hipExtLaunchKernelGGL((psm_encode_bf16_kernel<C, AL>), dim3(n_slices), dim3(256), (std::uint32_t)lds, st, ev_start, ev_stop, 0, a); \
  else PSM_LAUNCH((psm_encode_bf16_kernel<C, AL>), dim3(n_slices), dim3(256), lds, st, a)
#define ENC(C) case C: if (a.aligned) { ENC2(C, true); } else { ENC2(C, false); } break;
  switch (a.c_in) {
    ENC(1) ENC(2) ENC(3) ENC(4)
    default: return hipErrorInvalidValue;
  }
#undef ENC
#undef ENC2
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
template <int MTC>
__global__ __launch_bounds__(256) void psm_decode_bf16_kernel(PsmDecodeArgs a, int m_base) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int LDA = a.ld_res + 8, Q = a.ld_res / 4, G = a.ld_res / 16;   // ld_res is a multiple of 32
  const int ct = min(blockIdx.x * 4 + wave, a.n_coltiles - 1);
  const bool live = (blockIdx.x * 4 + wave) < a.n_coltiles;
  const bf16x8* bp = reinterpret_cast<const bf16x8*>(a.bpack) + ((int64_t)ct * G) * 64 + lane;
  // activation tile (f32 coefficients -> bf16) and per-row scales
  for (int idx = tid; idx < MTC * 32 * Q; idx += 256) {
    const int row = idx / Q, q = idx - row * Q;
    const int m = min(m_base + row, a.Mpad - 1);
    const v4f x = *reinterpret_cast<const v4f*>(a.res + (int64_t)m * a.ld_res + 4 * q);
    bf16x4 v;
    v[0] = (__bf16)x.x; v[1] = (__bf16)x.y; v[2] = (__bf16)x.z; v[3] = (__bf16)x.w;
    *reinterpret_cast<bf16x4*>(&ldsb[row * LDA + 4 * q]) = v;
  }
  float* lrs = reinterpret_cast<float*>(ldsb + MTC * 32 * LDA);
  if (tid < MTC * 32) lrs[tid] = a.row_scale[min(m_base + tid, a.Mpad - 1)];
  const int col = ct * 32 + i;
  const float mu = a.mean[col];
  __syncthreads();
  f32x16 acc[MTC];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) acc[mt] = (f32x16){0};
  for (int g0 = 0; g0 < G; g0 += 8) {            // 8 MFMA steps (128 k) of weights in flight
    bf16x8 b[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) b[g] = bp[(int64_t)min(g0 + g, G - 1) * 64];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      const __bf16* arow = &ldsb[(mt * 32 + i) * LDA + 8 * h + 16 * g0];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g0 + g < G) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + 16 * g);
          acc[mt] = MFMA_BF16(av, b[g], acc[mt]);
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int rr = mt * 32 + acc_row(rg, h);
      const int m = m_base + rr;
      if (m < a.M) a.pred[(int64_t)m * a.K_out + col] = (acc[mt][rg] + mu) * lrs[rr];
    }
  }
}

hipError_t psm_launch_decode_bf16(const PsmDecodeArgs& a, hipStream_t st) {
  const int nwg = (a.n_coltiles + 3) / 4;
  int m_base = 0;
  while (m_base < a.Mpad) {
    const int tiles = (a.Mpad - m_base) / 32;
    const int mtc = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const size_t lds = (size_t)mtc * 32 * (a.ld_res + 8) * 2 + (size_t)mtc * 32 * sizeof(float);
    if (mtc == 4) PSM_LAUNCH((psm_decode_bf16_kernel<4>), dim3(nwg), dim3(256), lds, st, a, m_base);
    else if (mtc == 2) PSM_LAUNCH((psm_decode_bf16_kernel<2>), dim3(nwg), dim3(256), lds, st, a, m_base);
    else PSM_LAUNCH((psm_decode_bf16_kernel<1>), dim3(nwg), dim3(256), lds, st, a, m_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    m_base += mtc * 32;
  }
  return hipSuccess;
}
