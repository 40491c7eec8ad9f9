// psm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of one surrogate solve.
//
//   encode  : split-K f32 MFMA GEMM  coeff = (blocks - mean) @ comp_in^T, the block
//             operand gathered straight from the grid image (blocks are never
//             materialised), centring fused into the LDS fill      [PM:303-349]
//   reduce  : split-K slab reduction + affine input scaler         [PM:351, SMD:505-523]
//   dense   : Keras Dense (x@W+b, ReLU / linear head + inverse scaler) [PM:121-134, SMD:532-539]
//   decode  : f32 MFMA GEMM  blocks = res @ comp_out + mean, out_scale fused [PM:365-366, SMD:541-551]
//   strips  : masked overlap-strip sums of the raw decoded blocks  [PM:391-445, SMD:233-316, UGP:300-340]
//   chain   : serial per-block offset recurrence + global shift    [same lines; PM:472, SMD:350, UGP:359-361]
//   paste   : owner-map gather of the corrected blocks into the field [PM:449-467, SMD:334-348, UGP:345-356]
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32 fma chain).  Operand maps (wave64):
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, reg r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// The K order inside a group of 8 is permuted (step j of group g uses k = 8g + 4h + j for
// lane half h) so that one 16-byte read per lane feeds four MFMAs; both operands use it.
#include "psm_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256) void psm_encode_kernel(PsmEncodeArgs a) {
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per workgroup
  constexpr int G = KS / 8;                     // groups of 8 k
  constexpr int LDA = KS + 4;                   // LDS row stride (floats): 16-B slots rotate by one per row
  constexpr int Q = KS / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  const int r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
  const int64_t src_off = (int64_t)r * a.row_stride + (int64_t)c0 * C_IN;
  const float* __restrict__ mean = a.mean + (int64_t)s * KS;
  const int NT = a.NT;
  const int i = lane & 31, h = lane >> 5;

  float4 b[G];
  int cur_t = -1;
  auto load_b = [&](int t) {
    const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
    for (int g = 0; g < G; ++g) b[g] = p[g * 64];
    cur_t = t;
  };
  if (wave < NT) load_b(wave);  // weights stream: issued before the activation tile is staged

  for (int m0 = 0; m0 < a.Mpad; m0 += 32 * PSM_MT_CHUNK) {
    const int rows = min(32 * PSM_MT_CHUNK, a.Mpad - m0);
    for (int idx = tid; idx < rows * Q; idx += 256) {
      const int row = idx / Q, q = idx - row * Q;
      const int m = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M) {
        const int64_t base = a.row_base[m];
        const float* src = a.grid + base + src_off + 4 * q;
        float4 x;
        if (ALIGNED) {
          x = *reinterpret_cast<const float4*>(src);
        } else {
          x = make_float4(src[0], src[1], src[2], src[3]);
        }
        const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * q);
        v = make_float4(x.x - mu.x, x.y - mu.y, x.z - mu.z, x.w - mu.w);
      }
      *reinterpret_cast<float4*>(&lds[row * LDA + 4 * q]) = v;
    }
    __syncthreads();
    for (int t = wave; t < NT; t += 4) {
      if (t != cur_t) load_b(t);
      for (int mt = 0; mt < rows / 32; ++mt) {
        f32x16 acc = {0};
        const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 av = *reinterpret_cast<const float4*>(arow + 8 * g);
          acc = MFMA32(av.x, b[g].x, acc);
          acc = MFMA32(av.y, b[g].y, acc);
          acc = MFMA32(av.z, b[g].z, acc);
          acc = MFMA32(av.w, b[g].w, acc);
        }
        float* out = a.part + ((int64_t)s * a.Mpad + m0 + mt * 32) * a.ldp + t * 32 + i;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg];
      }
    }
    __syncthreads();
  }
}

hipError_t psm_launch_encode(const PsmEncodeArgs& a, hipStream_t st) {
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  const int rows = a.Mpad < 32 * PSM_MT_CHUNK ? a.Mpad : 32 * PSM_MT_CHUNK;
  const size_t lds = (size_t)rows * (PSM_PIX_PER_SLICE * a.c_in + 4) * sizeof(float);
#define ENC(C)                                                                                   \
  case C:                                                                                        \
    if (a.aligned) hipLaunchKernelGGL((psm_encode_kernel<C, true>), dim3(n_slices), dim3(256), lds, st, a); \
    else hipLaunchKernelGGL((psm_encode_kernel<C, false>), dim3(n_slices), dim3(256), lds, st, a); \
    break;
  switch (a.c_in) {
    ENC(1) ENC(2) ENC(3) ENC(4)
    default: return hipErrorInvalidValue;
  }
#undef ENC
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// reduce (+ input scaler)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psm_reduce_kernel(PsmReduceArgs a) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)a.Mpad * a.ldp;
  const int64_t o = (int64_t)blockIdx.x * 64 + lane;
  const int per = (a.n_slices + 3) / 4;
  const int s0 = wave * per, s1 = min(a.n_slices, s0 + per);
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  const float* p = a.part + o;
  int s = s0;
  for (; s + 4 <= s1; s += 4) {
    acc0 += p[(int64_t)(s + 0) * total];
    acc1 += p[(int64_t)(s + 1) * total];
    acc2 += p[(int64_t)(s + 2) * total];
    acc3 += p[(int64_t)(s + 3) * total];
  }
  for (; s < s1; ++s) acc0 += p[(int64_t)s * total];
  red[wave][lane] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (wave == 0) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int col = (int)(o % a.ldp);
    a.xin[o] = v * a.ia[col] + a.ib[col];
  }
}

hipError_t psm_launch_reduce(const PsmReduceArgs& a, hipStream_t st) {
  const int64_t total = (int64_t)a.Mpad * a.ldp;
  hipLaunchKernelGGL(psm_reduce_kernel, dim3((unsigned)(total / 64)), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dense layer
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psm_dense_kernel(PsmDenseArgs a) {
  __shared__ float red[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockIdx.x, mt = blockIdx.y;
  const int i = lane & 31, h = lane >> 5;
  const int klen = a.Kpad / 4, kq = wave * klen;  // Kpad is a multiple of 32
  const float* arow = a.in + (int64_t)(mt * 32 + i) * a.ld_in + kq + 4 * h;
  const float* wcol = a.W + (int64_t)(kq + 4 * h) * a.ld_w + nt * 32 + i;
  f32x16 acc = {0};
  for (int g = 0; g < klen / 8; ++g) {
    const float4 av = *reinterpret_cast<const float4*>(arow + 8 * g);
    const float* w = wcol + (int64_t)(8 * g) * a.ld_w;
    const float b0 = w[0], b1 = w[a.ld_w], b2 = w[2 * (int64_t)a.ld_w], b3 = w[3 * (int64_t)a.ld_w];
    acc = MFMA32(av.x, b0, acc);
    acc = MFMA32(av.y, b1, acc);
    acc = MFMA32(av.z, b2, acc);
    acc = MFMA32(av.w, b3, acc);
  }
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) red[wave][acc_row(rg, h) * 33 + i] = acc[rg];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q, row = idx >> 5, col = idx & 31;
    const int n = nt * 32 + col;
    float v = (red[0][row * 33 + col] + red[1][row * 33 + col]) + (red[2][row * 33 + col] + red[3][row * 33 + col]);
    v += a.bias[n];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.head) v = v * a.sa[n] + a.sb[n];
    a.out[(int64_t)(mt * 32 + row) * a.ld_out + n] = v;
  }
}

hipError_t psm_launch_dense(const PsmDenseArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_dense_kernel, dim3(a.ld_w / 32, a.Mpad / 32), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
template <int MTC>
__global__ __launch_bounds__(256) void psm_decode_kernel(PsmDecodeArgs a, int m_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int LDA = a.ld_res + 4, Q = a.ld_res / 4;
  const int ct = blockIdx.x * 4 + wave;
  for (int idx = tid; idx < MTC * 32 * Q; idx += 256) {
    const int row = idx / Q, q = idx - row * Q;
    const int m = m_base + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.Mpad) v = *reinterpret_cast<const float4*>(a.res + (int64_t)m * a.ld_res + 4 * q);
    *reinterpret_cast<float4*>(&lds[row * LDA + 4 * q]) = v;
  }
  __syncthreads();
  if (ct >= a.n_coltiles) return;
  f32x16 acc[MTC];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) acc[mt] = (f32x16){0};
  const float4* bp = a.bpack + ((int64_t)ct * a.Gd) * 64 + lane;
  for (int g0 = 0; g0 < a.Gd; g0 += 4) {   // Gd is a multiple of 4
    float4 b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = bp[(int64_t)(g0 + g) * 64];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h + 8 * g0];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 av = *reinterpret_cast<const float4*>(arow + 8 * g);
        acc[mt] = MFMA32(av.x, b[g].x, acc[mt]);
        acc[mt] = MFMA32(av.y, b[g].y, acc[mt]);
        acc[mt] = MFMA32(av.z, b[g].z, acc[mt]);
        acc[mt] = MFMA32(av.w, b[g].w, acc[mt]);
      }
    }
  }
  const int col = ct * 32 + i;
  const float mu = a.mean[col];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int m = m_base + mt * 32 + acc_row(rg, h);
      if (m < a.M) a.pred[(int64_t)m * a.K_out + col] = (acc[mt][rg] + mu) * a.row_scale[m];
    }
  }
}

hipError_t psm_launch_decode(const PsmDecodeArgs& a, hipStream_t st) {
  const int nwg = (a.n_coltiles + 3) / 4;
  int m_base = 0;
  while (m_base < a.Mpad) {
    const int tiles = (a.Mpad - m_base) / 32;
    const int mtc = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const size_t lds = (size_t)mtc * 32 * (a.ld_res + 4) * sizeof(float);
    if (mtc == 4) hipLaunchKernelGGL((psm_decode_kernel<4>), dim3(nwg), dim3(256), lds, st, a, m_base);
    else if (mtc == 2) hipLaunchKernelGGL((psm_decode_kernel<2>), dim3(nwg), dim3(256), lds, st, a, m_base);
    else hipLaunchKernelGGL((psm_decode_kernel<1>), dim3(nwg), dim3(256), lds, st, a, m_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    m_base += mtc * 32;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// strips
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void psm_strips_kernel(PsmStripArgs a) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = blockIdx.x, f = blockIdx.y, cs = blockIdx.z;
  const int32_t* st = a.strips + (int64_t)e * 6;
  const int data = st[0], mask = st[1], r0 = st[2], r1 = st[3], c0 = st[4], c1 = st[5];
  const int w = c1 - c0, n = w * (r1 - r0);
  const int SS = a.S * a.S;
  const float* pred = a.pred + ((int64_t)(cs * a.B + data) * SS) * a.c_out + f;
  const float* gm = nullptr;
  if (mask >= 0)
    gm = a.grid + (((int64_t)cs * a.Ny + a.blk_y0x0[2 * mask]) * a.Nx + a.blk_y0x0[2 * mask + 1]) * a.c_in + a.sdf_ch;
  float sum = 0.f, cnt = 0.f;
  for (int idx = tid; idx < n; idx += 256) {
    const int rr = idx / w, cc = idx - rr * w;
    const int r = r0 + rr, c = c0 + cc;
    bool on = true;
    if (gm) on = gm[((int64_t)r * a.Nx + c) * a.c_in] != 0.f;
    if (on) {
      sum += pred[(int64_t)(r * a.S + c) * a.c_out];
      cnt += 1.f;
    }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = cnt; }
  __syncthreads();
  if (tid == 0) {
    const float S_ = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float C_ = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    a.sres[((int64_t)cs * a.c_out + f) * a.n_strips + e] = make_float2(S_, C_);
  }
}

hipError_t psm_launch_strips(const PsmStripArgs& a, int n_cases, hipStream_t st) {
  hipLaunchKernelGGL(psm_strips_kernel, dim3(a.n_strips, a.c_out, n_cases), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// chain (+ global shift)
// ---------------------------------------------------------------------------
struct PsmStripView {
  const float2* p;
  __device__ float mean(int s) const { const float2 v = p[s]; return v.x / v.y; }   // 0/0 -> NaN like np.mean([])
  __device__ float count(int s) const { return p[s].y; }
};

__global__ __launch_bounds__(64) void psm_chain_kernel(PsmChainArgs a) {
  extern __shared__ float sm[];   // [PSM_MAX_COLS] up + [B] offs
  float* up = sm;
  float* offs = sm + PSM_MAX_COLS;
  const int lane = threadIdx.x;
  const int f = blockIdx.x, cs = blockIdx.y;
  const int B = a.cp.B, SS = a.cp.S * a.cp.S;
  PsmStripView sv{a.sres + ((int64_t)cs * a.c_out + f) * a.n_strips};
  if (lane == 0) psm_chain<float>(a.cp, a.blocks, sv, f, up, offs);
  __syncthreads();
  float* go = a.offs + ((int64_t)cs * a.c_out + f) * B;
  for (int b = lane; b < B; b += 64) go[b] = offs[b];
  const int L = a.shiftL[f];
  const int32_t* la = a.shiftA + (int64_t)f * a.Lmax;
  const int32_t* lb = a.shiftB + (int64_t)f * a.Lmax;
  const float* pred = a.pred + ((int64_t)cs * B * SS) * a.c_out + f;
  float acc = 0.f;
  for (int k = lane; k < L; k += 64) {
    const int oa = a.owner[la[k]], ob = a.owner[lb[k]];
    const float va = oa >= 0 ? pred[(int64_t)oa * a.c_out] - offs[oa / SS] : 0.f;
    const float vb = ob >= 0 ? pred[(int64_t)ob * a.c_out] - offs[ob / SS] : 0.f;
    acc += 3.f * va - vb;
  }
  acc = wave_sum(acc);
  if (lane == 0) a.shift[cs * a.c_out + f] = acc / (float)L / 3.f;
}

hipError_t psm_launch_chain(const PsmChainArgs& a, int n_cases, hipStream_t st) {
  const size_t lds = (size_t)(PSM_MAX_COLS + a.cp.B) * sizeof(float);
  hipLaunchKernelGGL(psm_chain_kernel, dim3(a.c_out, n_cases), dim3(64), lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// paste
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psm_paste_kernel(PsmPasteArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int cs = blockIdx.y;
  if (pix >= a.npix) return;
  const int o = a.owner[pix];
  const int SS = a.S * a.S;
  float* out = a.fields + ((int64_t)cs * a.npix + pix) * a.c_out;
  if (o < 0) {
    for (int f = 0; f < a.c_out; ++f) out[f] = 0.f;
    return;
  }
  const int b = o / SS;
  const float* src = a.pred + ((int64_t)cs * a.B * SS + o) * a.c_out;
  for (int f = 0; f < a.c_out; ++f)
    out[f] = src[f] - a.offs[((int64_t)cs * a.c_out + f) * a.B + b] - a.shift[cs * a.c_out + f];
}

hipError_t psm_launch_paste(const PsmPasteArgs& a, int n_cases, hipStream_t st) {
  hipLaunchKernelGGL(psm_paste_kernel, dim3((a.npix + 255) / 256, n_cases), dim3(256), 0, st, a);
  return hipGetLastError();
}
