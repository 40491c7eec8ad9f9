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
#include "psm_devutil.h"
#include <type_traits>

#include <algorithm>
#include <cstdlib>

thread_local PsmLaunchProbe* psm_launch_probe = nullptr;

#include <hip/hip_ext.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Diagnostic stamps (100 MHz wall clock) -- compiled only with -DPSM_STAMPS, never in the shipped library.
#ifdef PSM_STAMPS
__device__ unsigned long long g_psm_stamps[64];
hipError_t psm_read_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psm_stamps), sizeof(g_psm_stamps)); }
#else
hipError_t psm_read_stamps(unsigned long long* out) { for (int i = 0; i < 64; ++i) out[i] = 0; return hipSuccess; }
#endif
#if defined(PSM_STAMPS) && !defined(PSM_STAMPS_ENC)      // -DPSM_STAMPS_ENC: all 64 slots belong to psm_encode_x6_mt_kernel (ESTAMP below)
#define PSM_STAMP(buf, k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_psm_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PSM_STAMP_T(tid_, k) do { if (blockIdx.x == 0 && threadIdx.x == (tid_) && (k) < 64) g_psm_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSM_STAMP(buf, k) do { } while (0)
#define PSM_STAMP_T(tid_, k) do { } while (0)
#endif

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Predicated 4-byte store without a branch: through a raw buffer descriptor over the whole destination, a lane that must not write
// gets the offset 0xffffffff, which the hardware's range check drops (round 6: the paste epilogues were sixteen s_and_saveexec /
// branch / 64-bit address / store sequences per row chunk -- half the instructions of a chunk of the batch decode).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t psm_store_rsrc(float* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void psm_store_if(__amdgpu_buffer_rsrc_t r, uint32_t elem, bool on, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, on ? elem * 4u : 0xffffffffu, 0, 0);
}

// streamed-once operands (PCA bases): -DPSM_NT_STREAM selects non-temporal loads (so that the 42 MB of basis data per
// solve do not displace the small tables and dense weights from the L2s).  Measured on MI355X: SLOWER, 44.1 vs
// 41.9 us per solve -- back-to-back solves re-read the bases from L2 / Infinity Cache, which nt gives up.  Off.
#ifdef PSM_NT_STREAM
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
#else
__device__ __forceinline__ float4 stream_load(const float4* p) { return *p; }
#endif

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256) void psm_encode_kernel(PsmEncodeArgs a) {
  psm_warm_kernargs<sizeof(PsmEncodeArgs)>();
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per workgroup
  constexpr int G = KS / 8;                     // groups of 8 k
  constexpr int LDA = KS + 4;                   // LDS row stride (floats): 16-B slots rotate by one per row
  constexpr int Q = KS / 4;                     // 16-byte pieces per activation row (<= 64)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  const int r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
  const int64_t src_off = (int64_t)r * a.row_stride + (int64_t)c0 * C_IN;
  const int NT = a.NT;
  const int i = lane & 31, h = lane >> 5;
  const int ql = lane < Q ? lane : Q - 1;       // lanes >= Q idle in the staging (C_IN < 4)
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + (int64_t)s * KS + 4 * ql);

  // One activation row per wave and step: the row's origin is wave-uniform (scalar load),
  // the lanes read 16 contiguous bytes each.  All loads of a batch are issued before any use.
  auto load_rows = [&](float4 (&x)[8], int m0, int row0) {    // rows row0 + wave + 4u (u < 8) of chunk m0
    int64_t rb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rb[u] = psm_row_base(a.row_base, min(m0 + row0 + wave + 4 * u, a.M - 1));   // wave-uniform: scalar loads
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* src = a.grid + rb[u] + src_off + 4 * ql;
      if (ALIGNED) x[u] = *reinterpret_cast<const float4*>(src);
      else x[u] = make_float4(src[0], src[1], src[2], src[3]);
    }
  };
  auto write_rows = [&](const float4 (&x)[8], int m0, int row0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = row0 + wave + 4 * u;
      const float keep = (m0 + row) < a.M ? 1.f : 0.f;          // padding rows -> 0 (no branch)
      const float4 v = make_float4((x[u].x - mu.x) * keep, (x[u].y - mu.y) * keep, (x[u].z - mu.z) * keep, (x[u].w - mu.w) * keep);
      if (lane < Q) *reinterpret_cast<float4*>(&lds[row * LDA + 4 * lane]) = v;
    }
  };
  auto stage_rows = [&](int m0, int row0) {
    float4 x[8];
    load_rows(x, m0, row0);
    write_rows(x, m0, row0);
  };
  auto gemm_tile = [&](const float4 (&b)[G], int mt, int t, int m0, bool store) {
    f32x16 acc = {0};
    const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h];
    float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < G ? g + 1 : g));   // next group in flight
      acc = MFMA32(av.x, b[g].x, acc);
      acc = MFMA32(av.y, b[g].y, acc);
      acc = MFMA32(av.z, b[g].z, acc);
      acc = MFMA32(av.w, b[g].w, acc);
      av = an;
    }
    if (store) {
      float* out = a.part + ((int64_t)s * a.Mpad + m0 + mt * 32) * a.ldp + t * 32 + i;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg];
    }
  };

  if (NT <= 4 && a.Mpad > 32 && a.Mpad <= 32 * PSM_MT_CHUNK && a.whole) {
    // 33..128 block rows (a per-GPU shard of a case batch: 8 cases x 9 blocks = 72 rows): ALL rows and the weight slice
    // are requested in one go -- one memory round trip in front of the MFMAs instead of one per 64-row chunk --, staged
    // (rows beyond M as zeros, always 128 of them: no conditional stores), then the 2-4 row tiles run back to back with
    // the partial-sum stores of tile mt under the MFMAs of tile mt + 1.
    // The first row tile and the weight slice are requested first; the first tile's MFMAs start as soon as its rows and
    // the first weight group have landed (counted vmcnt) and run under the rest of the stream; the other tiles are
    // staged after them (their own LDS rows: no hazard with tile 0 being read).
    const int t = min(wave, NT - 1);
    PSM_STAMP(0, 0);
    float4 x[PSM_MT_CHUNK][8];
    load_rows(x[0], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float4 b[G];
    {
      const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = stream_load(p + g * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    write_rows(x[0], 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PSM_STAMP(0, 1);
    // the other row tiles are requested only now: asked for up front, together with everything else, they delayed the
    // first tile's rows (5.3 us to the first MFMA instead of ~3); they land under the first tile's 3 us of MFMAs
#pragma unroll
    for (int q = 1; q < PSM_MT_CHUNK; ++q) load_rows(x[q], 0, 32 * q);
    __builtin_amdgcn_sched_barrier(0);
    gemm_tile(b, 0, t, 0, wave < NT);
    PSM_STAMP(0, 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 1; q < PSM_MT_CHUNK; ++q) write_rows(x[q], 0, 32 * q);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PSM_STAMP(0, 4);
    for (int mt = 1; mt < a.Mpad / 32; ++mt) gemm_tile(b, mt, t, 0, wave < NT);
    PSM_STAMP(0, 2);
    return;
  }

  if (NT <= 4 && a.Mpad > 32) {
    // many block rows (case batches): 64-row chunks double-buffered in LDS.  The rows of chunk c+1 are
    // requested before the MFMAs of chunk c and written to the other buffer after them, so the
    // staging round trip hides under the matrix work; the weight slice stays in registers throughout.
    // Barriers are LDS-only (the partial-sum stores need not drain between chunks).
    constexpr int CH = 64;
    // wave -> (component tile t, first row tile, row-tile step): with one or two component tiles (<= 64 components, e.g.
    // the reference's 45-component network) the waves split the two 32-row tiles of a chunk between them instead of
    // recomputing the last component tile (which halved the useful MFMA rate of this path)
    int t, mt_first, mt_step;
    bool store;
    if (NT == 2) { t = wave & 1; mt_first = wave >> 1; mt_step = 2; store = true; }
    else if (NT == 1) { t = 0; mt_first = wave & 1; mt_step = 2; store = wave < 2; }
    else { t = min(wave, NT - 1); mt_first = 0; mt_step = 1; store = wave < NT; }
    float4 xa[8], xb[8];
    load_rows(xa, 0, 0);
    load_rows(xb, 0, 32);
    __builtin_amdgcn_sched_barrier(0);
    float4 b[G];
    {
      const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = stream_load(p + g * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    write_rows(xa, 0, 0);
    write_rows(xb, 0, 32);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    for (int m0 = 0; m0 < a.Mpad; m0 += CH) {
      const bool more = m0 + CH < a.Mpad;
      if (more) { load_rows(xa, m0 + CH, 0); load_rows(xb, m0 + CH, 32); }
      const int tiles = min(2, (a.Mpad - m0) / 32);
      for (int mt = mt_first; mt < tiles; mt += mt_step) {
        f32x16 acc = {0};
        const float* arow = &lds[(buf * CH + mt * 32 + i) * LDA + 4 * h];
        float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < G ? g + 1 : g));
          acc = MFMA32(av.x, b[g].x, acc);
          acc = MFMA32(av.y, b[g].y, acc);
          acc = MFMA32(av.z, b[g].z, acc);
          acc = MFMA32(av.w, b[g].w, acc);
          av = an;
        }
        if (store) {
          float* out = a.part + ((int64_t)s * a.Mpad + m0 + mt * 32) * a.ldp + t * 32 + i;
#pragma unroll
          for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg];
        }
      }
      if (more) {
        // rows of the next chunk into the other buffer (row index relative to that buffer)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = wave + 4 * u;
          const float k0 = (m0 + CH + row) < a.M ? 1.f : 0.f, k1 = (m0 + CH + 32 + row) < a.M ? 1.f : 0.f;
          const float4 v0 = make_float4((xa[u].x - mu.x) * k0, (xa[u].y - mu.y) * k0, (xa[u].z - mu.z) * k0, (xa[u].w - mu.w) * k0);
          const float4 v1 = make_float4((xb[u].x - mu.x) * k1, (xb[u].y - mu.y) * k1, (xb[u].z - mu.z) * k1, (xb[u].w - mu.w) * k1);
          if (lane < Q) {
            *reinterpret_cast<float4*>(&lds[((buf ^ 1) * CH + row) * LDA + 4 * lane]) = v0;
            *reinterpret_cast<float4*>(&lds[((buf ^ 1) * CH + 32 + row) * LDA + 4 * lane]) = v1;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      buf ^= 1;
    }
    return;
  }

  if (NT <= 4 && a.Mpad <= 32 * PSM_MT_CHUNK) {
    // common case (<= 128 components, <= 128 block rows): straight-line so that the weight
    // stream stays in flight behind the first MFMAs (counted vmcnt).  Every wave computes (a
    // wave without a tile of its own recomputes the last one and stores nothing): keeps the
    // weight loads unconditional, ahead of the barrier.
    const int t = min(wave, NT - 1);
    PSM_STAMP(0, 0);
    float4 x0[8];
    load_rows(x0, 0, 0);                        // first 32 rows: loads issued BEFORE the weights
    __builtin_amdgcn_sched_barrier(0);
    float4 b[G];
    {
      const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = stream_load(p + g * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    write_rows(x0, 0, 0);                       // waits for the activation rows only (counted vmcnt)
    for (int row0 = 32; row0 < a.Mpad; row0 += 32) stage_rows(0, row0);
    __syncthreads();
    PSM_STAMP(0, 1);
    gemm_tile(b, 0, t, 0, wave < NT);           // peeled: counted waits on the weight stream
    for (int mt = 1; mt < a.Mpad / 32; ++mt) gemm_tile(b, mt, t, 0, wave < NT);
    PSM_STAMP(0, 2);
    return;
  }

  // general case: any number of component tiles / row chunks
  float4 b[G];
  int cur_t = -1;
  for (int m0 = 0; m0 < a.Mpad; m0 += 32 * PSM_MT_CHUNK) {
    const int rows = min(32 * PSM_MT_CHUNK, a.Mpad - m0);
    for (int row0 = 0; row0 < rows; row0 += 32) stage_rows(m0, row0);
    __syncthreads();
    for (int t = wave; t < NT; t += 4) {
      if (t != cur_t) {
        const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) b[g] = stream_load(p + g * 64);
        cur_t = t;
      }
      for (int mt = 0; mt < rows / 32; ++mt) gemm_tile(b, mt, t, m0, true);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// encode of ONE row tile (a single case, <= 32 block rows) with four component tiles: a workgroup owns TWO consecutive K
// slices and one HALF of the components -- waves (slice 0 | 1) x (component tile 2 half + (0 | 1)) -- and adds the two
// slices' partial sums through LDS before it stores: n_slices / 2 slabs instead of n_slices.  Same basis bytes, same
// MFMAs per wave (one slice x one component tile, the whole basis slice in registers, counted waits) as the one-slice
// form above; what halves is the slab traffic (4.2 -> 2.1 MB written) and, above all, what the ONE workgroup per block row
// of psm_reduce_dense1_kernel has to pull in front of the first Dense layer (128 -> 64 KB: that launch's longest phase).
// ---------------------------------------------------------------------------
template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256) void psm_encode_pair_kernel(PsmEncodeArgs a) {
  psm_warm_kernargs<sizeof(PsmEncodeArgs)>();
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per slice
  constexpr int G = KS / 8;                     // groups of 8 k
  constexpr int LDA = KS + 4;                   // LDS row stride (floats): 16-B slots rotate by one per row
  constexpr int Q = KS / 4;                     // 16-byte pieces per activation row (<= 64)
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // [2 slices][32 rows][LDA], then [2 tiles][64 lanes][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sl = wave >> 1, j = wave & 1;       // this wave's slice of the pair; its row parity in the staging and its component tile of the half
  const int pair = (int)blockIdx.x >> 1, half = (int)blockIdx.x & 1;
  const int s = 2 * pair + sl, t = 2 * half + j;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  const int r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
  const int64_t src_off = (int64_t)r * a.row_stride + (int64_t)c0 * C_IN;
  const int NT = a.NT;
  const int i = lane & 31, h = lane >> 5;
  const int ql = lane < Q ? lane : Q - 1;       // lanes >= Q idle in the staging (C_IN < 4)
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + (int64_t)s * KS + 4 * ql);
  // Request order (round 6): the block-row offsets are a dependent table lookup in front of the activation rows.  They are
  // wave-uniform, so they come through the scalar cache (address space 4: s_load_dwordx2, its own counter) instead of sixteen
  // wave-wide vector loads, and the FIRST HALF of the basis stream is requested before anything waits for them: the lookup's round
  // trip hides behind it.  Then the rows, then the second half of the stream; the staging below waits for the rows (and, the counter
  // being in order, the first half of the stream, which was requested earlier anyway), the MFMAs of k group g for groups <= g.
  float4 b[G];
  const float4* bp = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
  int64_t rb[16];
  {
#pragma unroll
    for (int u = 0; u < 16; ++u) rb[u] = psm_row_base(a.row_base, min(j + 2 * u, a.M - 1));              // wave-uniform
  }
  __builtin_amdgcn_sched_barrier(0);             // (the scheduler sinks the scalar loads behind the stream's first half otherwise)
#pragma unroll
  for (int g = 0; g < G / 2; ++g) b[g] = stream_load(bp + g * 64);
  __builtin_amdgcn_sched_barrier(0);
  // staging: the two waves of a slice take its even / odd rows (16 each); every request before any use
  float4 x[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const float* src = a.grid + rb[u] + src_off + 4 * ql;
    if (ALIGNED) x[u] = *reinterpret_cast<const float4*>(src);
    else x[u] = make_float4(src[0], src[1], src[2], src[3]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = G / 2; g < G; ++g) b[g] = stream_load(bp + g * 64);
  __builtin_amdgcn_sched_barrier(0);
  float* tile = lds + sl * 32 * LDA;
#pragma unroll
  for (int u = 0; u < 16; ++u) {                // waits for the activation rows only (counted vmcnt)
    const int row = j + 2 * u;
    const float keep = row < a.M ? 1.f : 0.f;   // padding rows -> 0 (no branch)
    const float4 v = make_float4((x[u].x - mu.x) * keep, (x[u].y - mu.y) * keep, (x[u].z - mu.z) * keep, (x[u].w - mu.w) * keep);
    if (lane < Q) *reinterpret_cast<float4*>(&tile[row * LDA + 4 * lane]) = v;
  }
  __syncthreads();
  f32x16 acc = {0};
  {
    const float* arow = &tile[i * LDA + 4 * h];
    float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < G ? g + 1 : g));   // next group in flight
      acc = MFMA32(av.x, b[g].x, acc);
      acc = MFMA32(av.y, b[g].y, acc);
      acc = MFMA32(av.z, b[g].z, acc);
      acc = MFMA32(av.w, b[g].w, acc);
      av = an;
    }
  }
  // slice 1's partial sums meet slice 0's through LDS ([tile j][register][lane]: conflict-free both ways); fixed order s0 + s1
  float* red = lds + 2 * 32 * LDA + j * 16 * 64;
  if (sl == 1) {
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) red[rg * 64 + lane] = acc[rg];
  }
  __syncthreads();
  if (sl == 0) {
    float* out = a.part + ((int64_t)pair * a.Mpad) * a.ldp + t * 32 + i;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg] + red[rg * 64 + lane];
  }
}

// ---------------------------------------------------------------------------
// encode, "x6" arithmetic: the same contraction on the bf16 matrix pipe at float32 accuracy.
// Every float32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significant bits: the two
// remainders x - hi and (x - hi) - mid are exact in float32), and a product a * w is taken as the six terms whose
// weight is >= 2^-16 of it -- hh, hm, mh, hl, lh, mm -- by v_mfma_f32_32x32x16_bf16 (products of bf16 pairs are exact in
// float32; accumulation in float32).  The three dropped terms are <= 2^-23 of the product, the size of one float32
// rounding.  Six MFMAs of 32 cycles cover 16 k where the float32 instruction (v_mfma_f32_32x32x2_f32, 64 cycles) needs
// eight: 192 vs 512 cycles, and the matrix phase is the longest serial phase of this kernel for case batches.
// The basis stays float32 in memory (same packing, same bytes): a wave splits its slice in registers while the rest of
// the stream is in flight; the activation rows are split once, on their way into LDS (three bf16 planes).
// k order inside a 16-step: lane half h holds k = 16 s + 4 h + (0..3) and 16 s + 8 + 4 h + (0..3) in both operands.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x6_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void psm_split3(f32x4 x, x6_bf16x4& h, x6_bf16x4& m, x6_bf16x4& l) {
  h = __builtin_convertvector(x, x6_bf16x4);                       // round to nearest even
  const f32x4 r1 = x - __builtin_convertvector(h, f32x4);          // exact
  m = __builtin_convertvector(r1, x6_bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);         // exact, <= 8 significant bits
  l = __builtin_convertvector(r2, x6_bf16x4);
}
__device__ __forceinline__ x6_bf16x8 psm_cat4(x6_bf16x4 a, x6_bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// LDS image of x6 activation planes read as ONE ds_read_b128 per MFMA operand: rows a multiple of 16 bytes (an odd number of 16-byte slots),
// the four 4-element groups of every 16 k stored 0, 2, 1, 3 (lane half h holds k 4h.. and 8 + 4h..).  Position (bf16) of 4-element group q:
__device__ __forceinline__ int psm_x6_group_pos(int q) { const int g = q & 3; return 16 * (q >> 2) + 4 * (g == 1 ? 2 : (g == 2 ? 1 : g)); }
#define MFMA_X6(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256) void psm_encode_x6_kernel(PsmEncodeArgs a) {
  psm_warm_kernargs<sizeof(PsmEncodeArgs)>();
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per workgroup
  constexpr int G = KS / 8;                     // float4 groups of the packed basis per lane
  constexpr int NS = KS / 16;                   // MFMA steps
  constexpr int LDB = KS + 4;                   // plane row stride in bf16: KS / 2 + 2 dwords = 2 * odd -> ds_read_b64 of 32 rows conflict-free
  constexpr int Q = KS / 4;
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsx[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  const int r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
  const int64_t src_off = (int64_t)r * a.row_stride + (int64_t)c0 * C_IN;
  const int NT = a.NT;
  const int i = lane & 31, h = lane >> 5;
  const int ql = lane < Q ? lane : Q - 1;
  // rows per plane (as the launcher sized the LDS).  33 ... 128 block rows run as TWO workgroups per slice (gridDim.y == 2),
  // 64 rows each: half the LDS, so two workgroups share a CU and one's matrix phase covers the other's load latency
  const int R = a.Mpad <= 32 ? 32 : (gridDim.y == 2 ? 64 : 32 * PSM_MT_CHUNK);
  const int PL = R * LDB;                                    // plane stride
  const int mrow0 = gridDim.y == 2 ? 64 * (int)blockIdx.y : 0;   // first block row of this workgroup
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + (int64_t)s * KS + 4 * ql);

  auto load_rows = [&](float4 (&x)[8], int m0, int row0) {
    int64_t rb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rb[u] = psm_row_base(a.row_base, min(m0 + row0 + wave + 4 * u, a.M - 1));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* src = a.grid + rb[u] + src_off + 4 * ql;
      if (ALIGNED) x[u] = *reinterpret_cast<const float4*>(src);
      else x[u] = make_float4(src[0], src[1], src[2], src[3]);
    }
  };
  auto write_rows = [&](const float4 (&x)[8], int m0, int row0, int buf_row0) {   // rows row0 + wave + 4u of chunk m0 -> LDS rows buf_row0 + ...
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = wave + 4 * u;
      const float keep = (m0 + row0 + row) < a.M ? 1.f : 0.f;
      const f32x4 v = {(x[u].x - mu.x) * keep, (x[u].y - mu.y) * keep, (x[u].z - mu.z) * keep, (x[u].w - mu.w) * keep};
      x6_bf16x4 vh, vm, vl;
      psm_split3(v, vh, vm, vl);
      if (lane < Q) {
        __bf16* dst = &ldsx[(buf_row0 + row) * LDB + 4 * lane];
        *reinterpret_cast<x6_bf16x4*>(dst) = vh;
        *reinterpret_cast<x6_bf16x4*>(dst + PL) = vm;
        *reinterpret_cast<x6_bf16x4*>(dst + 2 * PL) = vl;
      }
    }
  };
  x6_bf16x8 Bh[NS], Bm[NS], Bl[NS];
  // one 32-row tile against this wave's 32 components.  SPLIT: the basis registers b[] are split on the way (first tile:
  // each step waits only for the two groups it needs, the rest of the stream stays in flight)
  auto gemm_tile = [&](const float4 (&b)[G], bool split, int lds_row0, int out_row0, int t, bool store) {
    f32x16 acc = {0};
    const __bf16* arow = &ldsx[(lds_row0 + i) * LDB + 4 * h];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (split) {
        x6_bf16x4 h0, m0, l0, h1, m1, l1;
        psm_split3((f32x4){b[2 * st].x, b[2 * st].y, b[2 * st].z, b[2 * st].w}, h0, m0, l0);
        psm_split3((f32x4){b[2 * st + 1].x, b[2 * st + 1].y, b[2 * st + 1].z, b[2 * st + 1].w}, h1, m1, l1);
        Bh[st] = psm_cat4(h0, h1); Bm[st] = psm_cat4(m0, m1); Bl[st] = psm_cat4(l0, l1);
      }
      const x6_bf16x8 ah = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 16 * st + 8));
      const x6_bf16x8 am = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + PL + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + PL + 16 * st + 8));
      const x6_bf16x8 al = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 2 * PL + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 2 * PL + 16 * st + 8));
      acc = MFMA_X6(am, Bm[st], acc);            // small terms first
      acc = MFMA_X6(al, Bh[st], acc);
      acc = MFMA_X6(ah, Bl[st], acc);
      acc = MFMA_X6(am, Bh[st], acc);
      acc = MFMA_X6(ah, Bm[st], acc);
      acc = MFMA_X6(ah, Bh[st], acc);
    }
    if (store) {
      float* out = a.part + ((int64_t)s * a.Mpad + out_row0) * a.ldp + t * 32 + i;
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) out[(int64_t)acc_row(rg, h) * a.ldp] = acc[rg];
    }
  };
  auto load_basis = [&](float4 (&b)[G], int t) {
    const float4* p = a.bpack + (((int64_t)s * NT + t) * G) * 64 + lane;
#pragma unroll
    for (int g = 0; g < G; ++g) b[g] = stream_load(p + g * 64);
  };

  if (a.Mpad <= 32 * PSM_MT_CHUNK) {
    // up to 128 block rows: every row staged once; the first tile's rows and the basis slice are requested first, the
    // other tiles' rows land under the first tile's matrix work
    const int t = min(wave, NT - 1);
    const int tiles = min(a.Mpad - mrow0, R) / 32;           // row tiles of this workgroup (>= 1)
    float4 x0[8];
    load_rows(x0, mrow0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float4 b[G];
    load_basis(b, t);
    __builtin_amdgcn_sched_barrier(0);
    write_rows(x0, mrow0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float4 x1[8], x2[8], x3[8];
    if (tiles > 1) load_rows(x1, mrow0, 32);
    if (tiles > 2) load_rows(x2, mrow0, 64);
    if (tiles > 3) load_rows(x3, mrow0, 96);
    __builtin_amdgcn_sched_barrier(0);
    gemm_tile(b, true, 0, mrow0, t, wave < NT);
    if (tiles > 1) {
      __builtin_amdgcn_sched_barrier(0);
      write_rows(x1, mrow0, 32, 32);
      if (tiles > 2) write_rows(x2, mrow0, 64, 64);
      if (tiles > 3) write_rows(x3, mrow0, 96, 96);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      for (int mt = 1; mt < tiles; ++mt) gemm_tile(b, false, mt * 32, mrow0 + mt * 32, t, wave < NT);
    }
    return;
  }

  // many block rows: 64-row chunks double-buffered in LDS (rows of chunk c + 1 requested before the matrix work of chunk
  // c, written to the other buffer after it); with one or two component tiles the waves split the chunk's two row tiles
  constexpr int CH = 64;
  int t, mt_first, mt_step;
  bool store;
  if (NT == 2) { t = wave & 1; mt_first = wave >> 1; mt_step = 2; store = true; }
  else if (NT == 1) { t = 0; mt_first = wave & 1; mt_step = 2; store = wave < 2; }
  else { t = min(wave, NT - 1); mt_first = 0; mt_step = 1; store = wave < NT; }
  float4 xa[8], xb[8];
  load_rows(xa, 0, 0);
  load_rows(xb, 0, 32);
  __builtin_amdgcn_sched_barrier(0);
  float4 b[G];
  load_basis(b, t);
  __builtin_amdgcn_sched_barrier(0);
  write_rows(xa, 0, 0, 0);
  write_rows(xb, 0, 32, 32);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int buf = 0;
  bool first = true;
  for (int m0 = 0; m0 < a.Mpad; m0 += CH) {
    const bool more = m0 + CH < a.Mpad;
    if (more) { load_rows(xa, m0 + CH, 0); load_rows(xb, m0 + CH, 32); }
    const int tiles = min(2, (a.Mpad - m0) / 32);
    for (int mt = mt_first; mt < tiles; mt += mt_step) {
      if (first) gemm_tile(b, true, buf * CH + mt * 32, m0 + mt * 32, t, store);
      else gemm_tile(b, false, buf * CH + mt * 32, m0 + mt * 32, t, store);
      first = false;
    }
    if (more) {
      write_rows(xa, m0 + CH, 0, (buf ^ 1) * CH);
      write_rows(xb, m0 + CH, 32, (buf ^ 1) * CH + 32);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------
// encode, x6 arithmetic, M-TILED, for large case batches (>= 32 cases per step; BASELINE configs[3] on one card runs 64).
// psm_encode_x6_kernel gives every K slice of 192 its own workgroup, whatever the row count: at 576 block rows that is 256 slabs
// of 576 x 128 floats = 75 MB written by the launch and read again by the reduce.  Here a workgroup owns a GROUP of consecutive K
// slices x 64 block rows (two MFMA row tiles) x all components and keeps its accumulators in registers across the group: one slab
// per K group (56 groups at 64 cases: 16.5 MB), the activations are read exactly once, the basis once per row group -- from the
// XCD's L2 for all but the first (XCD-aware workgroup mapping below).
//   wave w = component tile w (32 components), as in psm_encode_x6_kernel.  Its basis comes PRE-SPLIT (three bf16 planes in MFMA
//   fragment order, psm_split_basis_kernel, 1.5 x the bytes): half a slice (96 k) at a time in 72 registers, loads and MFMAs only --
//   splitting the float32 pack in the kernel cost 344 vector instructions per half-slice and the registers for the raw copy.  The
//   activation rows of a (half-slice, row tile) step go through LDS (three bf16 planes, 32 rows, one buffer per row tile: 38 KB,
//   two workgroups per CU).  Round 6 (stamps: tools/encode_stamps.py; record: profiles/r06_case_batch.txt (6)): the next step's rows are
//   split and written to the OTHER tile's buffer in the shadow of the step's MFMA chain (sched_group_barrier: one MFMA, six vector
//   instructions, in turn) -- behind the chain the same instructions cost ~1 us of every 2.1 us step; rows are requested two steps
//   ahead, a half-slice's basis planes as the previous half-slice's second tile releases them, step by step.
// Measured at 64 cases (one box): 49 us + 6 us reduce against 60 + 12 us for the one-slab-per-slice form.  Built and measured on
// the way, none faster (profiles/r05_case_batch.txt): three row tiles with the float32 basis split in the kernel (49-56 us), an
// eight-wave form with four multiplying and four staging waves per workgroup (57-64 us: the staging wave of a SIMD runs its ~260
// instructions per step at half speed beside the multiplying wave), three workgroups per CU at 168 registers (spills: 64-82 us).
// Summation order: k ascending inside a K group (MFMA accumulators), then the groups in slab order (psm_reduce_kernel) --
// deterministic, but not the order of the one-slab-per-slice form (float32 rounding differs in the last bits).
// ---------------------------------------------------------------------------
#if defined(PSM_STAMPS) && defined(PSM_STAMPS_ENC)      // make stamps EXTRA=-DPSM_STAMPS_ENC: this kernel's stamps instead of the decode's (tools/encode_stamps.py)
#define ESTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == (PSM_STAMPS_ENC) && (k) < 64) g_psm_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ESTAMP(k) do { } while (0)
#endif
template <int C_IN, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void psm_encode_x6_mt_kernel(PsmEncodeArgs a) {
  psm_warm_kernargs<sizeof(PsmEncodeArgs)>();
  ESTAMP(0);
  constexpr int KS = PSM_PIX_PER_SLICE * C_IN;  // K elements per slice
  constexpr int KH = KS / 2;                    // ... per half-slice
  constexpr int NSH = KH / 16;                  // MFMA steps per half-slice
  constexpr int LDB = KH + 8;                   // plane row stride in bf16: a multiple of 16 bytes, an odd number of 16-byte slots (13 at C_in = 3)
                                                // -- the 16 lanes of a ds_read_b128 group fall on 16 different slots.  Inside every 16 k the four
                                                // 4-element groups are stored 0, 2, 1, 3: lane half h's MFMA operand (k 4h.. and 8 + 4h.., the order
                                                // of the basis pack) is ONE 16-byte read (as two 8-byte reads the compiler emitted ds_read2_b64:
                                                // 16 LDS cycles for what ds_read_b128 moves in 4, MI355X_MICROARCH.md LDS table)
  constexpr int QH = KH / 4;                    // float4 per activation row and half-slice
  constexpr int NX = (32 * QH + 255) / 256;     // float4 per thread and step
  constexpr int MT = PSM_ENC_MT_ROWS / 32;
  constexpr int PL = 32 * LDB;                  // plane stride (bf16)
  constexpr int MAXG = 8;                       // slices per K group, at most
  static_assert(KH % 16 == 0 && MT == 2, "whole MFMA steps per half-slice; two named row tiles");
  __shared__ __attribute__((aligned(16))) __bf16 ldsx[2 * 3 * PL];
  __shared__ __attribute__((aligned(16))) float mean_l[MAXG * KS];       // the workgroup's K range of the mean
  __shared__ int hs_off[2 * MAXG + 2];                                    // float offset of half-slice hs within a block row (read two ahead)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  // Workgroup -> (K group g, row group): the row groups of ONE K group read the same basis slices, so they get consecutive slots of
  // ONE XCD (workgroup i runs on XCD i % 8, tools/attic/xcc_probe.hip): id = ((g / 8) * row_groups + rg) * 8 + g % 8 -- the basis then comes
  // from memory once per K group and from that XCD's L2 for the other row groups.  Placement is a speed matter only.  Ids whose K
  // group does not exist (the last, partial block of eight) leave at once.
  const int n_groups = a.kgroup, row_groups = (a.Mpad + PSM_ENC_MT_ROWS - 1) / PSM_ENC_MT_ROWS;
  const int slot = blockIdx.x >> 3, g_blk = slot / row_groups, rg = slot - g_blk * row_groups;
  const int grp = g_blk * 8 + (blockIdx.x & 7);
  if (grp >= n_groups) return;
  const int s_first = (int)(((long long)grp * n_slices) / n_groups);
  const int s_end = (int)(((long long)(grp + 1) * n_slices) / n_groups);
  const int n_hs = 2 * (s_end - s_first);                    // half-slices of this workgroup (2 .. 2 MAXG)
  const int m0 = rg * PSM_ENC_MT_ROWS;
  const int runs = a.S / PSM_PIX_PER_SLICE;
  for (int k = tid; k < (s_end - s_first) * KS; k += 256) mean_l[k] = a.mean[(int64_t)s_first * KS + k];
  if (tid < n_hs) {
    const int s = s_first + (tid >> 1), r = s / runs, c0 = (s - r * runs) * PSM_PIX_PER_SLICE;
    hs_off[tid] = (int)((int64_t)r * a.row_stride + (int64_t)c0 * C_IN + (tid & 1) * KH);
  }
  const int NT = a.NT;
  const int t = min(wave, NT - 1);
  const int i = lane & 31, h = lane >> 5;
  // staging: 256 threads move one step's 32 rows x KH floats, float4 idx = tid + 256 u -> (row, q), fixed over the steps.  Everything a
  // request needs is ONE 32-bit float offset per (row tile, piece) -- row base (a case batch of grids is < 2^31 floats) + column --
  // plus the half-slice's scalar offset from LDS; padding rows are one bit each
  static_assert(32 * QH == 256 * NX, "every thread moves exactly NX pieces of a step");
  int ldst[NX], mq[NX];
  unsigned o0[NX], o1[NX];                                                 // BYTE offsets (the launcher keeps a case batch of grids under 4 GiB)
  unsigned keep_bits = 0;
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int idx = tid + 256 * u;
    const int xrow = idx / QH, xq = idx - xrow * QH;
    const int m = m0 + xrow;
    o0[u] = 4u * (unsigned)((int)a.row_base[min(m, a.M - 1)] + 4 * xq);
    o1[u] = 4u * (unsigned)((int)a.row_base[min(m + 32, a.M - 1)] + 4 * xq);
    ldst[u] = xrow * LDB + psm_x6_group_pos(xq);             // bf16 offset within a plane
    mq[u] = 4 * xq;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) keep_bits |= (m + 32 * mt) < a.M ? (1u << (mt * NX + u)) : 0u;
  }
  // one register set per row tile: a tile's rows are requested TWO steps ahead of their split (stamps, tools/encode_stamps.py: with one
  // step of lead the staging phase waited ~0.6 us per step for rows that come from HBM / MALL exactly once)
  float4 xr0[NX], xr1[NX];
  auto load_x = [&](int off, auto mt_tag) {                               // off = hs_off[hs] as a wave-uniform value: scalar base + 32-bit lane offset
    constexpr int MTI = decltype(mt_tag)::value;
    const char* base = reinterpret_cast<const char*>(a.grid + off);
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const float* src = reinterpret_cast<const float*>(base + (MTI == 0 ? o0[u] : o1[u]));
      float4& dst = MTI == 0 ? xr0[u] : xr1[u];
      if (ALIGNED) dst = *reinterpret_cast<const float4*>(src);
      else dst = make_float4(src[0], src[1], src[2], src[3]);
    }
  };
  float4 mu_r[NX];                                                         // the step's mean pieces, read at the top of its matrix phase
  auto read_mean = [&](int hs) {
#pragma unroll
    for (int u = 0; u < NX; ++u) mu_r[u] = *reinterpret_cast<const float4*>(&mean_l[hs * KH + mq[u]]);
  };
  auto write_piece = [&](auto mt_tag, int buf, int u) {
    constexpr int MTI = decltype(mt_tag)::value;
    const uint32_t k = ((keep_bits >> (MTI * NX + u)) & 1u) ? 0xffffffffu : 0u;   // padding rows: zeros (a bit mask: as a float factor the
    const float4 mu = mu_r[u];                                                     // compiler packed the multiplies -- v_pk_mul_f32 is slow beside MFMAs)
    const float4 x = MTI == 0 ? xr0[u] : xr1[u];
    const f32x4 v = {__uint_as_float(__float_as_uint(x.x - mu.x) & k), __uint_as_float(__float_as_uint(x.y - mu.y) & k),
                     __uint_as_float(__float_as_uint(x.z - mu.z) & k), __uint_as_float(__float_as_uint(x.w - mu.w) & k)};
    x6_bf16x4 vh, vm, vl;
    psm_split3(v, vh, vm, vl);
    __bf16* dst = &ldsx[buf * 3 * PL + ldst[u]];
    *reinterpret_cast<x6_bf16x4*>(dst) = vh;
    *reinterpret_cast<x6_bf16x4*>(dst + PL) = vm;
    *reinterpret_cast<x6_bf16x4*>(dst + 2 * PL) = vl;
  };
  auto write_x = [&](int hs, auto mt_tag, int buf) {
    read_mean(hs);
#pragma unroll
    for (int u = 0; u < NX; ++u) write_piece(mt_tag, buf, u);
  };
  // basis planes of a half-slice: [step][plane h, m, l] fragments as they lie in a.bpack_x6 (psm_split_basis_kernel): loads and
  // MFMAs only.  ONE register set (72), refilled step by step (mfma_tile's `next`)
  x6_bf16x8 P[NSH][3];
  auto b_ptr = [&](int hs) { return a.bpack_x6 + ((((int64_t)(s_first + (hs >> 1)) * NT + t) * (2 * NSH) + (hs & 1) * NSH) * 3) * 64 + lane; };
  auto load_b = [&](int hs) {
    const uint4* p = b_ptr(hs);
#pragma unroll
    for (int st = 0; st < NSH; ++st)
#pragma unroll
      for (int q = 0; q < 3; ++q) P[st][q] = __builtin_bit_cast(x6_bf16x8, p[(st * 3 + q) * 64]);
  };
  f32x16 acc0 = {0}, acc1 = {0};
  // `next` != nullptr: the planes of step st are dead after its six MFMAs in a half-slice's SECOND row tile -- the next half-slice's
  // planes of that step are requested right there (a whole matrix phase + staging ahead of their first use instead of behind the
  // last MFMA of the phase)
  // `stage(u)`: piece u of the NEXT step's rows (split + LDS writes into the other tile's buffer, which nobody reads during this step) rides
  // in the shadow of MFMA groups 2u, 2u + 1: one MFMA, then a few of its vector instructions, in turn (the MFMAs are one dependent chain of
  // 32 cycles each; behind the phase the same instructions cost ~1 us per step, tools/encode_stamps.py)
  auto mfma_tile = [&](f32x16& c, int buf, const uint4* next, int stamp, int hs_stage, auto&& stage) {
    const __bf16* arow = &ldsx[buf * 3 * PL + i * LDB + 8 * h];
    if (hs_stage >= 0) read_mean(hs_stage);
    x6_bf16x8 A[2][3];
    auto rd = [&](int st, int sl) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        A[sl][pl] = *reinterpret_cast<const x6_bf16x8*>(arow + pl * PL + 16 * st);
    };
    rd(0, 0);
#pragma unroll
    for (int st = 0; st < NSH; ++st) {
      const int sl = st & 1;
      if (st + 1 < NSH) rd(st + 1, sl ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if ((st & 1) == 0) stage(st >> 1);
      c = MFMA_X6(A[sl][1], P[st][1], c);            // small terms first: mm, lh, hl, mh, hm, hh
      c = MFMA_X6(A[sl][2], P[st][0], c);
      c = MFMA_X6(A[sl][0], P[st][2], c);
      c = MFMA_X6(A[sl][1], P[st][0], c);
      c = MFMA_X6(A[sl][0], P[st][1], c);
      c = MFMA_X6(A[sl][0], P[st][0], c);
#pragma unroll
      for (int k6 = 0; k6 < 6; ++k6) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      if (st == 0) ESTAMP(stamp);
      if (next) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) P[st][q] = __builtin_bit_cast(x6_bf16x8, next[(st * 3 + q) * 64]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  typedef std::integral_constant<int, 0> T0; typedef std::integral_constant<int, 1> T1;
  __syncthreads();                                            // mean_l, hs_off
  load_x(__builtin_amdgcn_readfirstlane(hs_off[0]), T0{});
  load_b(0);
  load_x(__builtin_amdgcn_readfirstlane(hs_off[0]), T1{});
  int off_next = __builtin_amdgcn_readfirstlane(hs_off[1]);    // hs_off[hs + 1] at the top of run_hs(hs); the one after is read a half-slice ahead
  __builtin_amdgcn_sched_barrier(0);
  write_x(0, T0{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  ESTAMP(1);
  // steps (half-slice hs, row tile): buffer = tile (two steps per half-slice).  The rows a step splits and writes beside its MFMAs (those
  // of the NEXT step) were requested before the PREVIOUS step's MFMAs; what it requests itself is for the step after next.  (The last
  // half-slice is a second compile-time copy: behind a run-time "is there a next one" the plane registers became a conditional
  // assignment and spilled.)
  auto run_hs = [&](int hs, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    if (more) load_x(off_next, T0{});
    const int off_raw = hs_off[hs + 2];
    __builtin_amdgcn_sched_barrier(0);
    mfma_tile(acc0, 0, nullptr, 2 + 6 * hs, hs, [&](int u) { write_piece(T1{}, 1, u); });
    ESTAMP(3 + 6 * hs);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ESTAMP(4 + 6 * hs);
    if (more) load_x(off_next, T1{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) mfma_tile(acc1, 1, b_ptr(hs + 1), 5 + 6 * hs, hs + 1, [&](int u) { write_piece(T0{}, 0, u); });
    else mfma_tile(acc1, 1, nullptr, 5 + 6 * hs, -1, [](int) {});
    off_next = __builtin_amdgcn_readfirstlane(off_raw);
    ESTAMP(6 + 6 * hs);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ESTAMP(7 + 6 * hs);
  };
  for (int hs = 0; hs + 1 < n_hs; ++hs) run_hs(hs, std::true_type{});
  run_hs(n_hs - 1, std::false_type{});
  if (wave < NT) {
    float* out = a.part + ((int64_t)grp * a.Mpad + m0) * a.ldp + t * 32 + i;
#pragma unroll
    for (int q = 0; q < 16; ++q) out[(int64_t)acc_row(q, h) * a.ldp] = acc0[q];
    if (m0 + 32 < a.Mpad) {
#pragma unroll
      for (int q = 0; q < 16; ++q) out[(int64_t)(32 + acc_row(q, h)) * a.ldp] = acc1[q];
    }
  }
  ESTAMP(63);
}

// x6 covers <= 128 components (NT <= 4) and an LDS footprint of three bf16 planes
static bool psm_encode_x6_fits(const PsmEncodeArgs& a, size_t* lds, int* row_wgs) {
  // Negative result, kept as a knob: two workgroups per slice (64 rows each, half the LDS, two per CU so that one's matrix
  // phase covers the other's load latency) -- 17.7 against 15.0 us at 8 cases, 21.8 against 17.6 us at 12: the launch is bound
  // by its streams, and the split reads the basis slice twice.  PSM_ENCODE_ROWSPLIT=1 selects it.
  static const bool row_split = getenv("PSM_ENCODE_ROWSPLIT") != nullptr;
  *row_wgs = (a.Mpad > 64 && a.Mpad <= 32 * PSM_MT_CHUNK && row_split) ? 2 : 1;
  const int rows = a.Mpad <= 32 ? 32 : (*row_wgs == 2 ? 64 : 32 * PSM_MT_CHUNK);
  *lds = (size_t)3 * rows * (PSM_PIX_PER_SLICE * a.c_in + 4) * 2;
  return a.NT <= 4 && *lds <= 156 * 1024 && a.Mpad % 32 == 0;
}

// the two-slices-per-workgroup form (psm_encode_pair_kernel): one row tile, exactly four component tiles, float32 MFMA, an even
// number of slices.  PSM_ENCODE_PAIRS=0 keeps one slab per slice.  The slab count the reduce launches must use: n_slices / 2 when psm_encode_pairs(args) (launch_all, psm_api_solve.cpp).
bool psm_encode_pairs(const PsmEncodeArgs& a) {
  static const bool on = !(getenv("PSM_ENCODE_PAIRS") && atoi(getenv("PSM_ENCODE_PAIRS")) == 0);
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  return on && a.pairs_ok && !a.x6 && a.kgroup <= 1 && a.Mpad == 32 && a.NT == 4 && n_slices % 2 == 0;
}
hipError_t psm_launch_encode(const PsmEncodeArgs& a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  const int rows = a.Mpad <= 32 ? 32 : 32 * PSM_MT_CHUNK;        // one tile, or 2 x 64-row buffers / a 128-row chunk
  size_t lds = (size_t)rows * (PSM_PIX_PER_SLICE * a.c_in + 4) * sizeof(float);
  size_t lds_x6 = 0;
  int row_wgs = 1;
  if (a.x6 && a.kgroup > 1) {
    // kgroup = number of K GROUPS here (the slab count); a group holds n_slices / kgroup slices, rounded either way, at most 8
    if (a.NT > 4 || a.Mpad % 32 != 0 || a.kgroup > n_slices || (n_slices + a.kgroup - 1) / a.kgroup > 8 || !a.bpack_x6) return hipErrorInvalidValue;
    const int row_groups = (a.Mpad + PSM_ENC_MT_ROWS - 1) / PSM_ENC_MT_ROWS;
    const dim3 grid((unsigned)(((a.kgroup + 7) / 8) * row_groups * 8));      // XCD-aware 1-D mapping, see the kernel
#define ENCM2(C, AL)                                                                                          \
  if (ev_start) hipExtLaunchKernelGGL((psm_encode_x6_mt_kernel<C, AL>), grid, dim3(256), 0, st, ev_start, ev_stop, 0, a); \
  else PSM_LAUNCH((psm_encode_x6_mt_kernel<C, AL>), grid, dim3(256), 0, st, a)
#define ENCM(C) case C: if (a.aligned) { ENCM2(C, true); } else { ENCM2(C, false); } break;
    switch (a.c_in) {
      ENCM(1) ENCM(2) ENCM(3) ENCM(4)
      default: return hipErrorInvalidValue;
    }
#undef ENCM
#undef ENCM2
    return hipGetLastError();
  }
  if (a.x6 && psm_encode_x6_fits(a, &lds_x6, &row_wgs)) {
    lds = lds_x6;
#define ENCX2(C, AL)                                                                                          \
  if (ev_start) hipExtLaunchKernelGGL((psm_encode_x6_kernel<C, AL>), dim3(n_slices, row_wgs), dim3(256), (std::uint32_t)lds, st, ev_start, ev_stop, 0, a); \
  else PSM_LAUNCH((psm_encode_x6_kernel<C, AL>), dim3(n_slices, row_wgs), dim3(256), lds, st, a)
#define ENCX(C) case C: if (a.aligned) { ENCX2(C, true); } else { ENCX2(C, false); } break;
    switch (a.c_in) {
      ENCX(1) ENCX(2) ENCX(3) ENCX(4)
      default: return hipErrorInvalidValue;
    }
#undef ENCX
#undef ENCX2
    return hipGetLastError();
  }
  if (psm_encode_pairs(a)) {                       // one row tile, four component tiles: two slices per workgroup, n_slices / 2 slabs
    const size_t lds_p = ((size_t)2 * 32 * (PSM_PIX_PER_SLICE * a.c_in + 4) + 2 * 16 * 64) * sizeof(float);
#define ENCP2(C, AL)                                                                                          \
  if (ev_start) hipExtLaunchKernelGGL((psm_encode_pair_kernel<C, AL>), dim3(n_slices), dim3(256), (std::uint32_t)lds_p, st, ev_start, ev_stop, 0, a); \
  else PSM_LAUNCH((psm_encode_pair_kernel<C, AL>), dim3(n_slices), dim3(256), lds_p, st, a)
#define ENCP(C) case C: if (a.aligned) { ENCP2(C, true); } else { ENCP2(C, false); } break;
    switch (a.c_in) {
      ENCP(1) ENCP(2) ENCP(3) ENCP(4)
      default: return hipErrorInvalidValue;
    }
#undef ENCP
#undef ENCP2
    return hipGetLastError();
  }
  // With events: hipExtLaunchKernelGGL stamps them with the dispatch's own begin / end times
  // (the source rocprofv3 reads), not with separate marker packets around the launch.
#define ENC2(C, AL)                                                                                          \
  if (ev_start) hipExtLaunchKernelGGL((psm_encode_kernel<C, AL>), dim3(n_slices), dim3(256), (std::uint32_t)lds, st, ev_start, ev_stop, 0, a); \
  else PSM_LAUNCH((psm_encode_kernel<C, AL>), dim3(n_slices), dim3(256), lds, st, a)
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
// basis of the large-batch encode: the float32 pack (pack_comp_in: [slice][ntile][KS/8 groups][64 lanes] float4) split exactly into
// three bf16 planes in the fragment order of the MFMA's second operand: [slice][ntile][KS/16 steps][plane h, m, l][64 lanes] x 8 bf16
// (step st = groups 2 st and 2 st + 1 of the same lane).  Once per handle (psm_api_solve.cpp, ensure_encode_aux).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psm_split_basis_kernel(const float4* bpack, uint4* out, long long n_frag, int NS) {
  const long long f = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // fragment = (slice, ntile, step)
  const int lane = threadIdx.x & 63;
  if (f >= n_frag) return;
  const long long tile = f / NS; const int st = (int)(f - tile * NS);
  const float4 g0 = bpack[(tile * (2 * NS) + 2 * st) * 64 + lane], g1 = bpack[(tile * (2 * NS) + 2 * st + 1) * 64 + lane];
  x6_bf16x4 h0, m0, l0, h1, m1, l1;
  psm_split3((f32x4){g0.x, g0.y, g0.z, g0.w}, h0, m0, l0);
  psm_split3((f32x4){g1.x, g1.y, g1.z, g1.w}, h1, m1, l1);
  out[(f * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, psm_cat4(h0, h1));
  out[(f * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, psm_cat4(m0, m1));
  out[(f * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, psm_cat4(l0, l1));
}
hipError_t psm_launch_split_basis(const float4* bpack, uint4* out, int n_slices, int NT, int KS, hipStream_t st) {
  if (KS % 16 != 0) return hipErrorInvalidValue;
  const long long n_frag = (long long)n_slices * NT * (KS / 16);
  PSM_LAUNCH(psm_split_basis_kernel, dim3((unsigned)((n_frag + 3) / 4)), dim3(256), 0, st, bpack, out, n_frag, KS / 16);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// reduce (+ input scaler): 16 waves x 16 slabs in flight per lane, one round trip
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void psm_reduce_kernel(PsmReduceArgs a) {
  __shared__ float red[16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)a.Mpad * a.ldp;
  const int64_t o = (int64_t)blockIdx.x * 64 + lane;
  const int per = (a.n_slices + 15) / 16;
  const int s0 = wave * per, s1 = min(a.n_slices, s0 + per);
  const float* p = a.part + o;
  PSM_STAMP(0, 16);
  float acc = 0.f;
  int s = s0;
  for (; s + 16 <= s1; s += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(s + u) * total];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];       // fixed order: deterministic
  }
  for (; s + 4 <= s1; s += 4) {                     // fewer than 16 slabs per wave (K groups, slice pairs): still every load of a batch in flight together
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[(int64_t)(s + u) * total];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  for (; s < s1; ++s) acc += p[(int64_t)s * total];
  red[wave][lane] = acc;
  __syncthreads();
  PSM_STAMP(0, 17);
  if (wave == 0) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][lane];
    const int col = (int)(o % a.ldp);
    a.xin[o] = v * a.ia[col] + a.ib[col];
  }
}

hipError_t psm_launch_reduce(const PsmReduceArgs& a, hipStream_t st) {
  const int64_t total = (int64_t)a.Mpad * a.ldp;
  PSM_LAUNCH(psm_reduce_kernel, dim3((unsigned)(total / 64)), dim3(1024), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// reduce + first dense layer in one launch (small row counts): one workgroup per block row
// (and column half of the layer) sums the row's split-K slabs, applies the input scaler and,
// having the WHOLE coefficient row in LDS, finishes x@W1+b1, ReLU for its columns on the
// VALU -- the contraction is only p_in (<= 512) long.  Saves a launch (~5 us) over
// psm_reduce_kernel + psm_dense_kernel; slab summation order is that of psm_reduce_kernel.
// ---------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(1024) void psm_reduce_dense1_kernel(PsmReduceArgs r, PsmDenseArgs d) {
  __shared__ float red[16][512];
  __shared__ __attribute__((aligned(16))) float xrow[512];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = blockIdx.x;
  const int64_t total = (int64_t)r.Mpad * r.ldp;
  const int per = (r.n_slices + 15) / 16;
  const int s0 = wave * per, s1 = min(r.n_slices, s0 + per);
  PSM_STAMP(0, 8);
  // first-layer weights of this thread's first column / K quarter: independent of the slabs, so
  // they are requested first and arrive under the slab reduction.  Addresses are a wave-uniform
  // row base plus a 32-bit lane offset.
  const int ncols = d.ld_w / gridDim.y, n0 = blockIdx.y * ncols;
  const int kp = wave >> 2, kq = d.Kpad / 4;               // 4 waves (256 columns) per K quarter
  const int nl0 = tid & 255;
  const int ncol0 = n0 + min(nl0, ncols - 1);
  float wv0[32];
  if (!BF16) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const float* wrow = d.W + (int64_t)(kp * kq + min(u, kq - 1)) * d.ld_w;     // uniform
      wv0[u] = wrow[ncol0];
    }
  }
  // input-scaler operands of the coefficient this thread finishes below
  const int pfin = min(tid, r.ldp - 1);
  const float ia_v = r.ia[pfin], ib_v = r.ib[pfin];
  const float bias0 = d.bias[n0 + min(tid, ncols - 1)];
  __builtin_amdgcn_sched_barrier(0);
  // slab sums: wave w adds slabs [16w, 16w+16) for two 64-column groups per pass, all 32 loads
  // of a pass in flight together (n_slices == 256: per == 16)
  for (int p0 = 0; p0 < r.ldp; p0 += 128) {
    const int pc0 = p0 + lane, pc1 = p0 + 64 + lane;
    const int o0 = m * r.ldp + min(pc0, r.ldp - 1), o1 = m * r.ldp + min(pc1, r.ldp - 1);
    float acc0 = 0.f, acc1 = 0.f;
    int s = s0;
    for (; s + 16 <= s1; s += 16) {
      float v0[16], v1[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* slab = r.part + (int64_t)(s + u) * total;                     // uniform
        v0[u] = slab[o0];
        v1[u] = slab[o1];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc0 += v0[u];     // fixed order: that of psm_reduce_kernel
#pragma unroll
      for (int u = 0; u < 16; ++u) acc1 += v1[u];
    }
    for (; s + 8 <= s1; s += 8) {                  // 128 slabs (slice pairs): 8 per wave, all 16 loads in flight together
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* slab = r.part + (int64_t)(s + u) * total;                     // uniform
        v0[u] = slab[o0];
        v1[u] = slab[o1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc0 += v0[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc1 += v1[u];
    }
    for (; s < s1; ++s) { const float* slab = r.part + (int64_t)s * total; acc0 += slab[o0]; acc1 += slab[o1]; }
    if (pc0 < r.ldp) red[wave][pc0] = acc0;
    if (pc1 < r.ldp) red[wave][pc1] = acc1;
  }
  __syncthreads();
  PSM_STAMP(0, 9);
  float x_keep = 0.f;
  if (tid < r.ldp) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][tid];
    x_keep = v * ia_v + ib_v;
    xrow[tid] = BF16 ? (float)(__bf16)x_keep : x_keep;
  }
  __syncthreads();
  // ---- x @ W1 + b1, ReLU: thread = (column, quarter of K); K = d.Kpad (multiple of 32)
  float* part4 = &red[0][0];                                           // [4][ncols <= 512]
  for (int nl = nl0; nl < ncols; nl += 256) {
    const int n = n0 + nl;
    float acc = 0.f;
    if (BF16) {
      const __bf16* w = reinterpret_cast<const __bf16*>(d.W) + (int64_t)(kp * kq) * d.ld_w + n;
      for (int k = 0; k < kq; ++k) acc = fmaf(xrow[kp * kq + k], (float)w[(int64_t)k * d.ld_w], acc);
    } else {
      // kq is a multiple of 8; up to 32 weight rows in flight per thread (one round trip for
      // p_in <= 128), k ascending.  Branch-free: rows beyond kq are clamped loads with a zero
      // weight, so that the LDS reads and FMAs of a chunk are one straight line.
      for (int k = 0; k < kq; k += 32) {
        float wv[32];
        if (k == 0 && nl == nl0) {
#pragma unroll
          for (int u = 0; u < 32; ++u) wv[u] = wv0[u];
        } else {
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const float* wrow = d.W + (int64_t)(kp * kq + min(k + u, kq - 1)) * d.ld_w;
            wv[u] = wrow[n];
          }
        }
        f32x4 xv[8];
#pragma unroll
        for (int u4 = 0; u4 < 8; ++u4)
          xv[u4] = *reinterpret_cast<const f32x4*>(&xrow[kp * kq + min(k + 4 * u4, kq - 4)]);
#pragma unroll
        for (int u4 = 0; u4 < 8; ++u4) {
          const bool in = (k + 4 * u4 < kq);                     // uniform; kq is a multiple of 4
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = fmaf(xv[u4][j], in ? wv[4 * u4 + j] : 0.f, acc);
        }
      }
    }
    part4[kp * 512 + nl] = acc;
  }
  __syncthreads();
  PSM_STAMP(0, 10);
  for (int nl = tid; nl < ncols; nl += 1024) {
    const int n = n0 + nl;
    float v = ((part4[nl] + part4[512 + nl]) + (part4[1024 + nl] + part4[1536 + nl])) + (nl == tid ? bias0 : d.bias[n]);
    if (d.relu) v = fmaxf(v, 0.f);
    if (d.head) v = v * d.sa[n] + d.sb[n];
    d.out[(int64_t)m * d.ld_out + n] = v;
  }
  // scaled coefficients, kept for psm_read_stage: stored last so that no barrier waits for them
  if (blockIdx.y == 0 && tid < r.ldp) r.xin[(int64_t)m * r.ldp + tid] = x_keep;
  PSM_STAMP(0, 11);
}

hipError_t psm_launch_reduce_dense1(const PsmReduceArgs& r, const PsmDenseArgs& d, hipStream_t st) {
  if (r.ldp > 512 || d.ld_w > 1024 || (d.ld_w / 2) % 1 != 0) return hipErrorInvalidValue;
  const dim3 grid(r.Mpad, 2);                        // (4 or 8 column workgroups per row: 4.64-4.76 us against 4.72 -- no difference)
  if (d.bf16) PSM_LAUNCH((psm_reduce_dense1_kernel<true>), grid, dim3(1024), 0, st, r, d);
  else PSM_LAUNCH((psm_reduce_dense1_kernel<false>), grid, dim3(1024), 0, st, r, d);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Conv1D over the PCA coefficients (conv1D_PCA head, NNs.py:75-124; the reference's 'conv1D' architecture has 7 layers of
// 128-64-32-16-32-64-128 filters, kernel 3, utils.py:452-454).  A rarely used head on <= a few hundred block rows of
// <= 128 positions: plain float32 FMAs, a thread owns 4 consecutive positions of one output channel (one weight load
// feeds 4 FMAs; lanes run over the output channels, so weight loads are coalesced and activation loads broadcast).
__global__ __launch_bounds__(256) void psm_conv1d_kernel(PsmConv1dArgs a) {
  const int m = blockIdx.y;
  const int items = ((a.P + 3) / 4) * a.c_out;
  const int item = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (item >= items) return;
  const int pt = item / a.c_out, co = item - pt * a.c_out;
  const int p0 = 4 * pt - (a.k - 1) / 2;                    // Keras 'same': (k - 1) / 2 zeros in front
  const float* in = a.in + (int64_t)m * a.in_stride;
  const float bv = a.bias[co];
  float acc[4] = {bv, bv, bv, bv};
  for (int t = 0; t < a.k; ++t) {
    const float* w = a.W + (int64_t)t * a.c_in * a.c_out + co;
    int pos[4]; float keep[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + t + j;
      keep[j] = (p >= 0 && p < a.P) ? 1.f : 0.f;
      pos[j] = min(max(p, 0), a.P - 1) * a.c_in;
    }
    for (int ci = 0; ci < a.c_in; ++ci) {
      const float wv = w[(int64_t)ci * a.c_out];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[pos[j] + ci] * keep[j], wv, acc[j]);
    }
  }
  float* out = a.out + (int64_t)m * a.out_stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = 4 * pt + j;
    if (p < a.P) out[(int64_t)p * a.c_out + co] = a.relu ? fmaxf(acc[j], 0.f) : acc[j];
  }
}

hipError_t psm_launch_conv1d(const PsmConv1dArgs& a, hipStream_t st) {
  if (a.M < 1 || a.P < 1 || a.k < 1 || a.k > 15 || a.c_in < 1 || a.c_out < 1) return hipErrorInvalidValue;
  const int items = ((a.P + 3) / 4) * a.c_out;
  PSM_LAUNCH(psm_conv1d_kernel, dim3((items + 255) / 256, a.M), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dense layer: v_mfma_f32_16x16x4_f32, one 16-column tile x 32 rows per workgroup,
// K split over 8 waves, operands prefetched to registers in one round trip.
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, reg r holds D[4*(l>>4) + r][l&15]
// k order inside a group of 16: step j uses k = 16g + 4*(l>>4) + j (one float4 of A per lane).
// ---------------------------------------------------------------------------
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// BF16: weights stored as bf16, activations rounded to bf16 on load; products are then exact
// and the f32 MFMA accumulates them exactly like v_mfma_*_bf16 would (this layer is latency
// bound, the bf16 storage only halves its weight bytes).
//
// Weights come MFMA-packed (psm_api_model.cpp, pack_dense): the four k of a lane's group are one
// 16-byte (bf16: 8-byte) piece and a wave's group is 1 KiB contiguous, so the whole operand set
// of a wave (NGC groups: 2*NGC + NGC loads per lane) is requested up front, unconditionally, and
// the MFMAs wait on it with counted vmcnt -- one memory round trip per pass.  Columns of the
// activation row beyond ld_in are clamped (their weights are zero rows).
// ROWS = 16 (few block rows: twice the workgroups, each pulling 2/3 of the bytes -- the layer
// is bound by what ONE CU can pull per round trip) or 32 (weights read once per 32 rows).
__device__ __forceinline__ float wave_sum(float v);
// A loaded value whose FIRST use would sit inside run-time predicated store blocks is consumed once before them, through an opaque
// move.  The wait-count pass cannot count the stores in flight behind run-time predicates, so with a load still pending at their
// first use it emits s_waitcnt vmcnt(0) in front of EVERY store: sixteen store round trips in series per epilogue (the decode
// kernels' `mean` value; tools/attic/isa_store_waits.py finds the pattern in a listing).
__device__ __forceinline__ float psm_settled(float v) {
  float r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
constexpr int PSM_DOTS_WG_ROWS = 256;   // closed-form dots: up to this many rows one workgroup per row, beyond it two rows per workgroup

// One guard wave (PsmGuardArgs): 8 ballots of 64 consecutive pixels each against the bound pattern.
__device__ __forceinline__ bool psm_guard_wave(const PsmGuardArgs& g, int gw, int lane) {
  constexpr int NB = PSM_GUARD_BALLOTS;
  float v[NB];
  unsigned long long want[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) {                       // all loads up front, clamped
    const long long pix = min((long long)(gw * NB + u) * 64 + lane, g.npix - 1);
    v[u] = g.sdf[pix * g.c_in];
    want[u] = g.bits[min(gw * NB + u, g.n_ballots - 1)];
  }
  bool bad = false;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const unsigned long long got = __ballot(v[u] != 0.f);           // NaN != 0 is true, like NumPy's `!= 0`
    bad |= (gw * NB + u < g.n_ballots) && got != want[u];
  }
  return bad;                                          // wave-uniform
}
// One guard workgroup (index gwg of this launch's range, uniform): PSM_GUARD_WG_WAVES guard waves' worth of pixels by the WAVES
// waves of the calling workgroup, one flag.  Every thread of the workgroup must call it (barriers).
template <int WAVES>
__device__ __forceinline__ void psm_guard_wg(const PsmGuardArgs& g, int gwg, int wave, int lane) {
  __shared__ int guard_bad[WAVES];
  if (gwg >= g.wg_count) return;                       // uniform
  const int wg = g.wg_first + gwg;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < PSM_GUARD_WG_WAVES / WAVES; ++k) {
    const int gw = wg * PSM_GUARD_WG_WAVES + wave * (PSM_GUARD_WG_WAVES / WAVES) + k;
    if (gw < g.n_waves) bad |= psm_guard_wave(g, gw, lane);
  }
  if (lane == 0) guard_bad[wave] = bad ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int any = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) any |= guard_bad[w];
    g.flags[wg] = any ? __int_as_float(0x7fc00000) : 0.f;
    if (any && g.host_flag) __hip_atomic_store(g.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// sum of the guard flags of a solve (0, or NaN after a mismatch): one wave, every lane gets the total
__device__ __forceinline__ float psm_guard_part(const float* flags, int n, int lane, int first) {   // this lane's share from `first` on
  float gs = 0.f;
  for (int k0 = first + lane; k0 < n; k0 += 64 * 8) {  // 8 independent loads per round
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = flags[min(k0 + 64 * u, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) gs += (k0 + 64 * u < n) ? v[u] : 0.f;
  }
  return gs;
}
__device__ __forceinline__ float psm_guard_sum(const float* flags, int n, int lane, float f0, float f1) {
  const float gs = (lane < n ? f0 : 0.f) + (lane + 64 < n ? f1 : 0.f) + psm_guard_part(flags, n, lane, 128);
  return wave_sum(gs);
}

// DOTS (head layer of the geometry-bound path): workgroups with blockIdx.z > 0 do not compute the layer but the
// strip dot products of psm_kernels.h (PsmDotsArgs) from the same input activation: one wave per two table rows,
// every load issued up front (clamped), out[row] = scale * (act . g2[row] + c2[row]) / cnt[row]  (0/0 = NaN for an
// empty strip, like np.mean([])).
// LNIN (hidden layers of densePCA_attention): the input carries a pending LayerNormalization (PsmDenseArgs::ln_*) -- moments of
// the workgroup's own rows in a prologue, operands normalised on their way into the MFMAs, optional residual in the epilogue.
// float4 number q (columns 4 q .. 4 q + 3) of row `row` of a Dense input: rows of ld_in floats, or the packed form of PsmDenseArgs
__device__ __forceinline__ f32x4 psm_act_q(const PsmDenseArgs& a, int row, int q) {
  if (a.in_rows) return reinterpret_cast<const f32x4*>(a.in_rows + (int64_t)row * a.ld_in)[q];
  if (a.in_packed) return reinterpret_cast<const f32x4*>(a.in)[((int64_t)(row >> 4) * (a.ld_in >> 4) + (q >> 2)) * 64 + (q & 3) * 16 + (row & 15)];
  return reinterpret_cast<const f32x4*>(a.in + (int64_t)row * a.ld_in)[q];
}

template <int NGC, bool BF16, int ROWS, bool DOTS, bool LNIN = false>   // NGC: groups of 16 k per wave per pass
__global__ __launch_bounds__(512) void psm_dense_kernel(PsmDenseArgs a, PsmDotsArgs d) {
  psm_warm_kernargs<sizeof(PsmDenseArgs) + sizeof(PsmDotsArgs)>();
  // (One column tile of 16 per workgroup.  Two, for 576 block rows x 512 columns = 576 workgroups on 512 slots, measured no gain: 7.7 / 8.7 us
  // either way -- the layer is 4.3 us of latency + 147 456 float32 MFMAs of 32 cycles on 1024 SIMDs; profiles/r05_case_batch.txt (6a).)
  if (!DOTS && blockIdx.z > 0) {                       // guard riders behind a hidden layer (large case batches)
    const int wg = ((int)(blockIdx.z - 1) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    psm_guard_wg<8>(d.guard, wg, threadIdx.x >> 6, threadIdx.x & 63);
    return;
  }
  if (DOTS && blockIdx.z > 0) {
    constexpr int RPW = 2, NQ = 4;                     // rows per wave; float4 per lane and row (Kh <= 1024)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = ((int)(blockIdx.z - 1) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    const bool wave_rows = d.n_src > 1 && d.n_rows > PSM_DOTS_WG_ROWS;      // closed form, many rows: two rows per workgroup
    const int n_dot_wgs = d.n_src > 1 ? (wave_rows ? (d.n_rows + 1) / 2 : d.n_rows) : (d.n_rows + 8 * RPW - 1) / (8 * RPW);
    if (wg >= n_dot_wgs) {                               // guard riders behind the dots workgroups (uniform per workgroup)
      psm_guard_wg<8>(d.guard, wg - n_dot_wgs, wave, lane);
      return;
    }
    const int nq = d.Kh / 4;
    if (wave_rows) {
      // closed form, case batches of more than PSM_DOTS_WG_ROWS rows: TWO rows per workgroup -- four waves per row, wave q of a
      // row takes the source blocks q, q + 4, ... (four per batch, all loads of a batch up front; up to 16 blocks are one round
      // trip), the partial sums meet in LDS.  64 cases x 9 rows: 288 workgroups instead of 576 behind the head's 144.
      __shared__ float lsum2[8];
      const int rsel = wave >> 2, q4 = wave & 3;
      const int row = wg * 2 + rsel, rc = min(row, d.n_rows - 1);
      const int cs = rc / d.rows_per_case;
      const f32x4* gp = reinterpret_cast<const f32x4*>(d.g2) + (int64_t)rc * d.n_src * nq;
      const int arow_base = cs * d.n_src;
      const float rsv = d.row_scale[cs * d.n_src];
      float acc = 0.f;
      for (int b0 = q4; b0 < d.n_src; b0 += 16) {
        f32x4 gg[4][NQ], xx[4][NQ];
        float cc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int blk = min(b0 + 4 * t, d.n_src - 1);
          cc[t] = d.c2[(int64_t)rc * d.n_src + blk];
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const int q = min(lane + 64 * u, nq - 1);
            gg[t][u] = gp[(int64_t)blk * nq + q];
            xx[t][u] = psm_act_q(a, arow_base + blk, q);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool on = b0 + 4 * t < d.n_src;
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const float s4 = (gg[t][u].x * xx[t][u].x + gg[t][u].y * xx[t][u].y) + (gg[t][u].z * xx[t][u].z + gg[t][u].w * xx[t][u].w);
            acc += (on && lane + 64 * u < nq) ? s4 : 0.f;
          }
          acc += (on && lane == 0) ? cc[t] : 0.f;
        }
      }
      const float tot = wave_sum(acc);
      if (lane == 0) lsum2[wave] = tot;
      __syncthreads();
      if (lane == 0 && q4 == 0 && row < d.n_rows) d.out[row] = rsv * ((lsum2[4 * rsel] + lsum2[4 * rsel + 1]) + (lsum2[4 * rsel + 2] + lsum2[4 * rsel + 3]));
      return;
    }
    if (d.n_src > 1) {
      // closed form: one WORKGROUP per row -- wave w takes the source blocks w, w + 8, ... (four per batch, all loads of a
      // batch up front), the eight partial sums meet in LDS
      __shared__ float lsum[8];
      const int row = wg, rc = min(row, d.n_rows - 1);
      const int cs = rc / d.rows_per_case;
      const f32x4* gp = reinterpret_cast<const f32x4*>(d.g2) + (int64_t)rc * d.n_src * nq;
      const int arow_base = cs * d.n_src;
      const float rsv = d.row_scale[cs * d.n_src];
      float acc = 0.f;
      for (int b0 = wave; b0 < d.n_src; b0 += 32) {
        f32x4 gg[4][NQ], xx[4][NQ];
        float cc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int blk = min(b0 + 8 * t, d.n_src - 1);
          cc[t] = d.c2[(int64_t)rc * d.n_src + blk];
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const int q = min(lane + 64 * u, nq - 1);
            gg[t][u] = gp[(int64_t)blk * nq + q];
            xx[t][u] = psm_act_q(a, arow_base + blk, q);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool on = b0 + 8 * t < d.n_src;
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const float s4 = (gg[t][u].x * xx[t][u].x + gg[t][u].y * xx[t][u].y) + (gg[t][u].z * xx[t][u].z + gg[t][u].w * xx[t][u].w);
            acc += (on && lane + 64 * u < nq) ? s4 : 0.f;
          }
          acc += (on && lane == 0) ? cc[t] : 0.f;
        }
      }
      const float tot = wave_sum(acc);
      if (lane == 0) lsum[wave] = tot;
      __syncthreads();
      if (threadIdx.x == 0 && row < d.n_rows) {
        float t8 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) t8 += lsum[w8];
        d.out[row] = rsv * t8;
      }
      return;
    }
    f32x4 g[RPW][NQ], x[RPW][NQ];
    float c2[RPW], cn[RPW], rs[RPW];
    int row[RPW];
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      row[t] = (wg * 8 + wave) * RPW + t;
      const int rc = min(row[t], d.n_rows - 1);
      const int blk = d.row_of[rc];
      c2[t] = d.c2[rc]; cn[t] = d.cnt[rc]; rs[t] = d.row_scale[blk];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int q = min(lane + 64 * u, nq - 1);
        g[t][u] = reinterpret_cast<const f32x4*>(d.g2)[(int64_t)rc * nq + q];
        x[t][u] = psm_act_q(a, blk, q);
      }
    }
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const float s4 = (g[t][u].x * x[t][u].x + g[t][u].y * x[t][u].y) + (g[t][u].z * x[t][u].z + g[t][u].w * x[t][u].w);
        acc += (lane + 64 * u < nq) ? s4 : 0.f;
      }
      const float tot = wave_sum(acc);
      if (lane == 0 && row[t] < d.n_rows) d.out[row[t]] = rs[t] * (tot + c2[t]) / cn[t];
    }
    return;
  }
  __shared__ float red[8][2][16 * 17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x, mt = blockIdx.y;
  const int i = lane & 15, kq = lane >> 4;
  const int groups = a.Kp / 16;                      // all waves
  const int ng = groups / 8;                         // per wave: a multiple of NGC
  PSM_STAMP(0, 44 + 4 * (a.layer & 3));
  // epilogue operands of this thread's output column: in flight from the start
  const int n_out = nt * 16 + (tid & 15);
  const float bias_v = a.bias[n_out];
  const float sa_v = a.head ? a.sa[n_out] : 1.f, sb_v = a.head ? a.sb[n_out] : 0.f;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const float* arow0 = a.in + (int64_t)(mt * ROWS + i) * a.ld_in;
  const float* arow1 = arow0 + (int64_t)16 * a.ld_in;
  const int g_first = wave * ng;
  const int kmax = a.ld_in - 4;
  auto rnd = [](float v) { return BF16 ? (float)(__bf16)v : v; };
  // ---- operand loads, normalisation and matrix step of NGC groups (one pass of the contraction)
  auto load_a = [&](int g0, f32x4 (&a0)[NGC], f32x4 (&a1)[NGC]) {
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int kcol = min(16 * (g_first + g0 + g) + 4 * kq, kmax);
      a0[g] = *reinterpret_cast<const f32x4*>(arow0 + kcol);
      if (ROWS == 32) a1[g] = *reinterpret_cast<const f32x4*>(arow1 + kcol);
    }
  };
  auto load_gb = [&](int g0, f32x4 (&gm)[NGC], f32x4 (&bt)[NGC]) {       // columns beyond ln_n carry gamma = beta = 0 (and zero weight rows)
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int kcol = min(16 * (g_first + g0 + g) + 4 * kq, kmax);
      gm[g] = *reinterpret_cast<const f32x4*>(a.ln_gamma + kcol);
      bt[g] = *reinterpret_cast<const f32x4*>(a.ln_beta + kcol);
    }
  };
  auto load_w = [&](int g0, f32x4 (&w)[NGC]) {
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int64_t widx = ((int64_t)nt * groups + g_first + g0 + g) * 64 + lane;
      if (BF16) {
        const uint2 u = reinterpret_cast<const uint2*>(a.Wp)[widx];
        w[g] = (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
      } else {
        w[g] = reinterpret_cast<const f32x4*>(a.Wp)[widx];
      }
    }
  };
  auto load_aw = [&](int g0, f32x4 (&a0)[NGC], f32x4 (&a1)[NGC], f32x4 (&w)[NGC]) {
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      const int kcol = min(16 * (g_first + g0 + g) + 4 * kq, kmax);
      if (!LNIN && !BF16 && ROWS == 32 && a.in_packed) {                 // uniform: one contiguous KiB per wave and row tile
        const int gin = a.ld_in >> 4;
        const f32x4* pk = reinterpret_cast<const f32x4*>(a.in) + ((int64_t)(2 * mt) * gin + min(g_first + g0 + g, gin - 1)) * 64 + lane;
        a0[g] = pk[0];
        a1[g] = pk[(int64_t)gin * 64];
      } else {
        a0[g] = *reinterpret_cast<const f32x4*>(arow0 + kcol);
        if (ROWS == 32) a1[g] = *reinterpret_cast<const f32x4*>(arow1 + kcol);
      }
      const int64_t widx = ((int64_t)nt * groups + g_first + g0 + g) * 64 + lane;
      if (BF16) {
        const uint2 u = reinterpret_cast<const uint2*>(a.Wp)[widx];
        w[g] = (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
      } else {
        w[g] = reinterpret_cast<const f32x4*>(a.Wp)[widx];
      }
      __builtin_amdgcn_sched_barrier(0);               // keeps the groups' requests in this order (the scheduler clusters the row loads otherwise)
    }
  };
  auto mma = [&](const f32x4 (&a0)[NGC], const f32x4 (&a1)[NGC], const f32x4 (&w)[NGC]) {
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      acc0 = MFMA16(rnd(a0[g].x), w[g].x, acc0); if (ROWS == 32) acc1 = MFMA16(rnd(a1[g].x), w[g].x, acc1);
      acc0 = MFMA16(rnd(a0[g].y), w[g].y, acc0); if (ROWS == 32) acc1 = MFMA16(rnd(a1[g].y), w[g].y, acc1);
      acc0 = MFMA16(rnd(a0[g].z), w[g].z, acc0); if (ROWS == 32) acc1 = MFMA16(rnd(a1[g].z), w[g].z, acc1);
      acc0 = MFMA16(rnd(a0[g].w), w[g].w, acc0); if (ROWS == 32) acc1 = MFMA16(rnd(a1[g].w), w[g].w, acc1);
    }
  };
  // ---- pending LayerNormalization of the input: moments of rows i (and i + 16) over the first ln_n columns, two passes like
  // tf.nn.moments.  Lane (i, kq) of wave w owns columns 16 (g_first + g) + 4 kq + j; the four kq lanes of a row meet through
  // two shuffles, the eight waves through LDS (the `red` buffer, free until the MFMA results are written).  A contraction of
  // one pass (K <= 512: ng == NGC) takes the moments from its operand registers -- the row is loaded once, with the weights
  // and gamma / beta already in flight; longer rows are read from L2 again for each pass.
  __shared__ float ln_stat[2][32];
  float mean0 = 0.f, rstd0 = 1.f, mean1 = 0.f, rstd1 = 1.f, res_raw = 0.f, res_g = 0.f, res_b = 0.f;
  f32x4 p0[NGC], p1[NGC], pw[NGC], pg[NGC], pb[NGC];
  const bool single = LNIN && ng == NGC;                 // uniform
  if constexpr (LNIN) {
    if (single) { load_a(0, p0, p1); load_gb(0, pg, pb); load_w(0, pw); }
    if (a.ln_residual && tid < ROWS * 16) {              // the epilogue's residual operands: in flight from here
      const int row = tid >> 4;
      res_raw = a.in[(int64_t)(mt * ROWS + row) * a.ld_in + n_out];
      res_g = a.ln_gamma[n_out]; res_b = a.ln_beta[n_out];
    }
    const float inv_n = 1.f / (float)a.ln_n;
    auto meet = [&](float s0, float s1, float& o0, float& o1) {
      s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      if (kq == 0) { red[wave][0][i] = s0; red[wave][1][i] = s1; }
      __syncthreads();
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) { t0 += red[w8][0][i]; t1 += red[w8][1][i]; }
      __syncthreads();                                   // everyone has read the partials: `red` may be rewritten
      o0 = t0 * inv_n; o1 = t1 * inv_n;
    };
    auto row_moment = [&](float m0, float m1, bool second, float& o0, float& o1) {
      float s0 = 0.f, s1 = 0.f;
      if (single) {
#pragma unroll
        for (int g = 0; g < NGC; ++g) {
          const int k0 = 16 * (g_first + g) + 4 * kq;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool in = k0 + j < a.ln_n;
            const float d0 = p0[g][j] - m0, d1 = (ROWS == 32 ? p1[g][j] : p0[g][j]) - m1;
            s0 += in ? (second ? d0 * d0 : d0) : 0.f;
            s1 += in ? (second ? d1 * d1 : d1) : 0.f;
          }
        }
      } else {
        for (int g = 0; g < ng; ++g) {
          const int k0 = 16 * (g_first + g) + 4 * kq, kc = min(k0, kmax);
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(arow0 + kc);
          f32x4 v1 = v0;
          if (ROWS == 32) v1 = *reinterpret_cast<const f32x4*>(arow1 + kc);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool in = k0 + j < a.ln_n;
            const float d0 = v0[j] - m0, d1 = v1[j] - m1;
            s0 += in ? (second ? d0 * d0 : d0) : 0.f;
            s1 += in ? (second ? d1 * d1 : d1) : 0.f;
          }
        }
      }
      meet(s0, s1, o0, o1);
    };
    float q0, q1;
    row_moment(0.f, 0.f, false, mean0, mean1);
    row_moment(mean0, mean1, true, q0, q1);
    rstd0 = rsqrtf(q0 + a.ln_eps); rstd1 = rsqrtf(q1 + a.ln_eps);
    if (wave == 0 && kq == 0) {
      ln_stat[0][i] = mean0; ln_stat[1][i] = rstd0;
      if (ROWS == 32) { ln_stat[0][16 + i] = mean1; ln_stat[1][16 + i] = rstd1; }
    }
  }
  auto normalise = [&](f32x4 (&a0)[NGC], f32x4 (&a1)[NGC], const f32x4 (&gm)[NGC], const f32x4 (&bt)[NGC]) {
#pragma unroll
    for (int g = 0; g < NGC; ++g) {
      a0[g] = (a0[g] - mean0) * rstd0 * gm[g] + bt[g];
      if (ROWS == 32) a1[g] = (a1[g] - mean1) * rstd1 * gm[g] + bt[g];
    }
  };
  if (single) {
    normalise(p0, p1, pg, pb);
    mma(p0, p1, pw);
  } else {
    for (int g0 = 0; g0 < ng; g0 += NGC) {
      f32x4 a0[NGC], a1[NGC], w[NGC];
      if constexpr (LNIN) {
        f32x4 gm[NGC], bt[NGC];
        load_a(g0, a0, a1);
        load_gb(g0, gm, bt);
        load_w(g0, w);
        normalise(a0, a1, gm, bt);
      } else {
        // requests in the order the MFMAs consume them -- (rows, weights) of k group 0, then of group 1, ...: loads return in issue
        // order, so the first group's MFMAs need vmcnt(3 (NGC - 1)) instead of everything but the last weight groups, and the
        // matrix steps of group g run while groups g + 1 ... are still arriving (round 6; was: all rows, then all weights)
        load_aw(g0, a0, a1, w);
      }
      __builtin_amdgcn_sched_barrier(0);               // every request of the pass before its first MFMA (the scheduler had sunk half of them behind it: two round trips)
      mma(a0, a1, w);
    }
  }
  PSM_STAMP(0, 45 + 4 * (a.layer & 3));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][0][(4 * kq + r) * 17 + i] = acc0[r];
    if (ROWS == 32) red[wave][1][(4 * kq + r) * 17 + i] = acc1[r];
  }
  __syncthreads();
  if (tid < ROWS * 16) {
    const int row = tid >> 4, col = tid & 15;          // ROWS rows x 16 cols
    const int half = row >> 4, r16 = row & 15;
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) v += red[w8][half][r16 * 17 + col];
    v += bias_v;
    if (a.relu) v = fmaxf(v, 0.f);
    if constexpr (LNIN) {
      if (a.ln_residual) v += (res_raw - ln_stat[0][row]) * ln_stat[1][row] * res_g + res_b;     // x + LN(input) (NNs.py:64)
    }
    if (a.head) v = v * sa_v + sb_v;
    if (!LNIN && !BF16 && ROWS == 32 && a.out_packed) {
      a.out[psm_packed_offset(mt * ROWS + row, n_out, a.ld_out >> 4)] = v;
      if (a.out_rows) a.out_rows[(int64_t)(mt * ROWS + row) * a.ld_out + n_out] = v;
    } else a.out[(int64_t)(mt * ROWS + row) * a.ld_out + n_out] = v;
  }
  PSM_STAMP(0, 46 + 4 * (a.layer & 3));
}

hipError_t psm_launch_dense(const PsmDenseArgs& a, hipStream_t st, const PsmGuardArgs* riders) {
  const int ng = a.Kp / 128;                       // groups of 16 k per wave
  if (!a.Wp || a.Kp % 128 != 0 || (ng > 2 && ng % 4 != 0) || a.ld_in < 4) return hipErrorInvalidValue;
  const bool r16 = a.Mpad <= 128;     // up to 128 block rows: 16-row tiles keep >= 64 workgroups pulling <= 64 KB each
  if ((a.in_packed || a.out_packed) && (r16 || a.bf16 || a.ln_gamma || (a.in_packed && (a.ld_in % 16 != 0 || a.Kp > a.ld_in)) || (a.out_packed && a.ld_out % 16 != 0)))
    return hipErrorInvalidValue;       // packed activations: float32 layers on 32-row tiles, no pending LayerNormalization
  const int gx = a.ld_w / 16, gy = a.Mpad / (r16 ? 16 : 32);
  PsmDotsArgs rd{};
  int gz = 1;
  if (riders && riders->sdf && riders->wg_count > 0) { rd.guard = *riders; gz = 1 + (riders->wg_count + gx * gy - 1) / (gx * gy); }
  const dim3 grid(gx, gy, gz), blk(512);
#define DENSE2(N, L)                                                                        \
  do {                                                                                      \
    if (r16) {                                                                              \
      if (a.bf16) PSM_LAUNCH((psm_dense_kernel<N, true, 16, false, L>), grid, blk, 0, st, a, rd); \
      else PSM_LAUNCH((psm_dense_kernel<N, false, 16, false, L>), grid, blk, 0, st, a, rd);       \
    } else {                                                                                \
      if (a.bf16) PSM_LAUNCH((psm_dense_kernel<N, true, 32, false, L>), grid, blk, 0, st, a, rd); \
      else PSM_LAUNCH((psm_dense_kernel<N, false, 32, false, L>), grid, blk, 0, st, a, rd);       \
    }                                                                                       \
  } while (0)
#define DENSE(N) do { if (a.ln_gamma) DENSE2(N, true); else DENSE2(N, false); } while (0)
  if (psm_launch_probe) psm_launch_probe->tag = a.layer;          // every Dense layer is its own entry of psm_time_kernels
  if (a.ln_gamma && (!a.ln_beta || a.ln_n < 1 || a.ln_n > a.ld_in || (a.ln_residual && a.ln_n > a.ld_w))) return hipErrorInvalidValue;
  if (ng == 1) DENSE(1); else if (ng == 2) DENSE(2); else DENSE(4);
  if (psm_launch_probe) psm_launch_probe->tag = -1;
#undef DENSE
#undef DENSE2
  return hipGetLastError();
}

hipError_t psm_launch_dense_dots(const PsmDenseArgs& a, const PsmDotsArgs& d, hipStream_t st) {
  const int ng = a.Kp / 128;
  if (!a.Wp || a.Kp % 128 != 0 || (ng > 2 && ng % 4 != 0) || a.ld_in < 4 || a.bf16) return hipErrorInvalidValue;
  if (d.Kh < 4 || d.Kh % 4 != 0 || d.Kh > 1024 || d.Kh > a.ld_in || d.n_rows < 1) return hipErrorInvalidValue;
  const bool r16 = a.Mpad <= 128;                   // same tile choice as psm_launch_dense
  if (a.out_packed || (a.in_packed && (r16 || a.ld_in % 16 != 0 || a.Kp > a.ld_in))) return hipErrorInvalidValue;
  const int gx = a.ld_w / 16, gy = a.Mpad / (r16 ? 16 : 32);
  // z planes > 0: ceil(n_rows / 16) dots workgroups (8 waves x 2 rows), then ceil(guard waves / 8) guard workgroups
  const int extra = (d.n_src > 1 ? (d.n_rows > PSM_DOTS_WG_ROWS ? (d.n_rows + 1) / 2 : d.n_rows) : (d.n_rows + 15) / 16) + (d.guard.sdf ? d.guard.wg_count : 0);
  const dim3 grid(gx, gy, 1 + (extra + gx * gy - 1) / (gx * gy)), blk(512);
#define DD(N)                                                                                          \
  do {                                                                                                 \
    if (r16) PSM_LAUNCH((psm_dense_kernel<N, false, 16, true>), grid, blk, 0, st, a, d);       \
    else PSM_LAUNCH((psm_dense_kernel<N, false, 32, true>), grid, blk, 0, st, a, d);           \
  } while (0)
  if (psm_launch_probe) psm_launch_probe->tag = a.layer;
  if (ng == 1) DD(1); else if (ng == 2) DD(2); else DD(4);
  if (psm_launch_probe) psm_launch_probe->tag = -1;
#undef DD
  return hipGetLastError();
}

// LayerNormalization (+ residual) of the densePCA_attention stack, see psm_kernels.h.  Two-pass moments like
// tf.nn.moments (mean, then the mean of squared deviations; biased variance), float32.
// NPL > 0: the row (n <= 64 * NPL) is read ONCE into NPL registers per lane, every load issued up front, and both moments come
// from the registers (one memory round trip); NPL == 0: any n, three passes over a row that sits in L2.
template <int NPL>
__global__ __launch_bounds__(256) void psm_layernorm_kernel(PsmLayerNormArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = (int)blockIdx.x * 4 + wave;
  if (row >= a.rows) return;                          // wave-uniform
  float* x = a.act + (int64_t)row * a.ld_act;
  const float* r = a.res ? a.res + (int64_t)row * a.ld_res : nullptr;
  const float inv_n = 1.f / (float)a.n;
  if constexpr (NPL > 0) {
    float v[NPL], rv[NPL], g[NPL], b[NPL];
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      const int k = min(lane + 64 * u, a.n - 1);
      v[u] = x[k]; rv[u] = r ? r[k] : 0.f; g[u] = a.gamma[k]; b[u] = a.beta[k];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NPL; ++u) { v[u] += rv[u]; s += (lane + 64 * u < a.n) ? v[u] : 0.f; }
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NPL; ++u) { const float d = v[u] - mean; q += (lane + 64 * u < a.n) ? d * d : 0.f; }
    const float inv = rsqrtf(wave_sum(q) * inv_n + a.eps);
#pragma unroll
    for (int u = 0; u < NPL; ++u)
      if (lane + 64 * u < a.n) x[lane + 64 * u] = (v[u] - mean) * inv * g[u] + b[u];
    return;
  }
  float s = 0.f;
  for (int k = lane; k < a.n; k += 64) s += x[k] + (r ? r[k] : 0.f);
  const float mean = wave_sum(s) * inv_n;
  float q = 0.f;
  for (int k = lane; k < a.n; k += 64) { const float d = x[k] + (r ? r[k] : 0.f) - mean; q += d * d; }
  const float inv = rsqrtf(wave_sum(q) * inv_n + a.eps);
  for (int k = lane; k < a.n; k += 64) x[k] = (x[k] + (r ? r[k] : 0.f) - mean) * inv * a.gamma[k] + a.beta[k];
}

hipError_t psm_launch_layernorm(const PsmLayerNormArgs& a, hipStream_t st) {
  if (!a.act || !a.gamma || !a.beta || a.rows < 1 || a.n < 1 || a.n > a.ld_act || (a.res && a.n > a.ld_res)) return hipErrorInvalidValue;
  const dim3 grid((a.rows + 3) / 4), blk(256);
  if (a.n <= 512) PSM_LAUNCH(psm_layernorm_kernel<8>, grid, blk, 0, st, a);
  else if (a.n <= 1024) PSM_LAUNCH(psm_layernorm_kernel<16>, grid, blk, 0, st, a);
  else PSM_LAUNCH(psm_layernorm_kernel<0>, grid, blk, 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
// Fast path: <= 128 output components (ld_res == 128, 16 groups of 8 k): the whole weight
// slice of a wave (16 x 16 B per lane) is issued behind the activation-tile loads and stays in
// flight under the first MFMAs (counted vmcnt).
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MTC>
__global__ __launch_bounds__(256) void psm_decode128_kernel(PsmDecodeArgs a, int m_first, int m_end) {
  constexpr int LDR = 128, LDA = LDR + 4, Q = LDR / 4, GD = LDR / 8, NA = MTC * 32 * Q / 256;
  m_first += (int)blockIdx.y * MTC * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int ct = min(blockIdx.x * 4 + wave, a.n_coltiles - 1);
  const bool live = (blockIdx.x * 4 + wave) < a.n_coltiles;
  auto load_tile = [&](v4f (&x)[NA], int m_base) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, row = idx / Q, q = idx - row * Q;
      const int m = min(m_base + row, a.Mpad - 1);
      x[u] = *reinterpret_cast<const v4f*>(a.res + (int64_t)m * LDR + 4 * q);
    }
  };
  auto write_tile = [&](const v4f (&x)[NA]) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, row = idx / Q, q = idx - row * Q;
      *reinterpret_cast<v4f*>(&lds[row * LDA + 4 * q]) = x[u];
    }
  };
  PSM_STAMP(0, 20);
  v4f x[NA];
  load_tile(x, m_first);
  float rs = a.row_scale[min(m_first + min(tid, MTC * 32 - 1), a.Mpad - 1)];
  __builtin_amdgcn_sched_barrier(0);
  float4 b[GD];                                  // this wave's weight slice: loaded once, kept for every row chunk
  const float4* bp = a.bpack + ((int64_t)ct * GD) * 64 + lane;
#pragma unroll
  for (int g = 0; g < GD; ++g) b[g] = stream_load(bp + g * 64);
  const int col = ct * 32 + i;
  const float mu_raw = a.mean[col];
  __builtin_amdgcn_sched_barrier(0);
  const float mu = psm_settled(mu_raw);
  float* lrs = lds + MTC * 32 * LDA;             // [MTC*32] out_scale per block row
  const int m_step = MTC * 32 * (int)gridDim.y;    // row chunks are dealt round-robin to the gridDim.y row groups
  for (int m_base = m_first; m_base < m_end; m_base += m_step) {
    if (m_base != m_first) {                     // later chunks (many block rows): only the activation tile is new
      __syncthreads();                           // every wave is done with the previous tile
      load_tile(x, m_base);
      rs = a.row_scale[min(m_base + min(tid, MTC * 32 - 1), a.Mpad - 1)];
    }
    write_tile(x);
    if (tid < MTC * 32) lrs[tid] = rs;
    __syncthreads();
    PSM_STAMP(0, 21);
    f32x16 acc[MTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      acc[mt] = (f32x16){0};
      const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h];
      float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
      for (int g = 0; g < GD; ++g) {
        const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < GD ? g + 1 : g));
        acc[mt] = MFMA32(av.x, b[g].x, acc[mt]);
        acc[mt] = MFMA32(av.y, b[g].y, acc[mt]);
        acc[mt] = MFMA32(av.z, b[g].z, acc[mt]);
        acc[mt] = MFMA32(av.w, b[g].w, acc[mt]);
        av = an;
      }
    }
    PSM_STAMP(0, 22);
    if (live) {
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
  }
  PSM_STAMP(0, 23);
}

template <int MTC, int GCH>   // GCH: groups of 8 k whose weights are prefetched together
__global__ __launch_bounds__(256) void psm_decode_kernel(PsmDecodeArgs a, int m_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int LDA = a.ld_res + 4, Q = a.ld_res / 4;
  const int ct = min(blockIdx.x * 4 + wave, a.n_coltiles - 1);
  const bool live = (blockIdx.x * 4 + wave) < a.n_coltiles;
  const float4* bp = a.bpack + ((int64_t)ct * a.Gd) * 64 + lane;
  // activation tile first (needed first), then the weight stream
  for (int idx = tid; idx < MTC * 32 * Q; idx += 256) {
    const int row = idx / Q, q = idx - row * Q;
    const int m = m_base + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.Mpad) v = *reinterpret_cast<const float4*>(a.res + (int64_t)m * a.ld_res + 4 * q);
    *reinterpret_cast<float4*>(&lds[row * LDA + 4 * q]) = v;
  }
  float4 b[GCH];
#pragma unroll
  for (int g = 0; g < GCH; ++g) b[g] = bp[(int64_t)min(g, a.Gd - 1) * 64];
  const int col = ct * 32 + i;
  const float mu = a.mean[col];
  __syncthreads();
  f32x16 acc[MTC];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) acc[mt] = (f32x16){0};
  for (int g0 = 0; g0 < a.Gd; g0 += GCH) {   // Gd is a multiple of 4; GCH in {4, 16}
    if (g0 > 0) {
#pragma unroll
      for (int g = 0; g < GCH; ++g) b[g] = bp[(int64_t)min(g0 + g, a.Gd - 1) * 64];
    }
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h + 8 * g0];
#pragma unroll
      for (int g = 0; g < GCH; ++g) {
        if (g0 + g < a.Gd) {
          const float4 av = *reinterpret_cast<const float4*>(arow + 8 * g);
          acc[mt] = MFMA32(av.x, b[g].x, acc[mt]);
          acc[mt] = MFMA32(av.y, b[g].y, acc[mt]);
          acc[mt] = MFMA32(av.z, b[g].z, acc[mt]);
          acc[mt] = MFMA32(av.w, b[g].w, acc[mt]);
        }
      }
    }
  }
  if (!live) return;
  const float mu_r = psm_settled(mu);
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
    float rsv[16];                                   // the tile's row scales first, straight-line (see psm_settled)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) rsv[rg] = a.row_scale[min(m_base + mt * 32 + acc_row(rg, h), a.M - 1)];
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) rsv[rg] = psm_settled(rsv[rg]);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int m = m_base + mt * 32 + acc_row(rg, h);
      if (m < a.M) a.pred[(int64_t)m * a.K_out + col] = (acc[mt][rg] + mu_r) * rsv[rg];
    }
  }
}

hipError_t psm_launch_decode(const PsmDecodeArgs& a, hipStream_t st) {
  const int nwg = (a.n_coltiles + 3) / 4;
  if (a.ld_res == 128) {
    // <= 128 components: ONE launch for any number of block rows; a workgroup keeps its weight slice in
    // registers and walks the rows in chunks of MTC*32 (chunk size chosen to waste the fewest padded tiles)
    // one output channel gives only 128 column workgroups: the row tiles are then spread over up to 512 / nwg row groups,
    // one chunk each where possible, so that all 256 CUs work (each group re-reads the weight slice: 8 MB more traffic
    // per group; same rule as psm_launch_decode_paste_batch, where it was measured)
    const int tiles = a.Mpad / 32;
    int mtc, groups;
    if (nwg >= 256) {
      const int iters = (tiles + 3) / 4;
      mtc = (tiles + iters - 1) / iters; groups = 1;
    } else {
      const int cap = std::max(1, 512 / nwg);
      mtc = std::min(4, std::max(1, (tiles + cap - 1) / cap));
      groups = std::min((tiles + mtc - 1) / mtc, cap);
    }
    const size_t lds128 = (size_t)mtc * 32 * (a.ld_res + 4) * sizeof(float) + (size_t)mtc * 32 * sizeof(float);
    const dim3 grid(nwg, groups);
    if (mtc == 4) PSM_LAUNCH((psm_decode128_kernel<4>), grid, dim3(256), lds128, st, a, 0, a.Mpad);
    else if (mtc == 3) PSM_LAUNCH((psm_decode128_kernel<3>), grid, dim3(256), lds128, st, a, 0, a.Mpad);
    else if (mtc == 2) PSM_LAUNCH((psm_decode128_kernel<2>), grid, dim3(256), lds128, st, a, 0, a.Mpad);
    else PSM_LAUNCH((psm_decode128_kernel<1>), grid, dim3(256), lds128, st, a, 0, a.Mpad);
    return hipGetLastError();
  }
  int m_base = 0;
  while (m_base < a.Mpad) {
    const int tiles = (a.Mpad - m_base) / 32;
    const int mtc = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const size_t lds = (size_t)mtc * 32 * (a.ld_res + 4) * sizeof(float);
    const bool g16 = (a.Gd % 16 == 0);
#define DEC(M_, G_) PSM_LAUNCH((psm_decode_kernel<M_, G_>), dim3(nwg), dim3(256), lds, st, a, m_base)
    if (mtc == 4) { if (g16) DEC(4, 16); else DEC(4, 4); }
    else if (mtc == 2) { if (g16) DEC(2, 16); else DEC(2, 4); }
    else { if (g16) DEC(1, 16); else DEC(1, 4); }
#undef DEC
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    m_base += mtc * 32;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// strips: one workgroup per (block, band of 16 rows).  Every decoded value of the band
// (the block's own and the previous block's, the latter under THIS block's mask) is read
// once into registers; the band's contribution to each of the block's strip rectangles is
// reduced over the workgroup and stored as a partial (sum per field, count).
// ---------------------------------------------------------------------------
// 64-lane sum on the VALU (DPP row shifts + row broadcasts, ~6 instructions) instead of
// __shfl_down, which lowers to ds_bpermute (an LDS round trip per step).  Every lane must be
// active; the total is returned in all lanes (readlane 63).
__device__ __forceinline__ float wave_sum(float v) {
#define PSM_DPP_ADD(ctrl, rmask)                                                                             \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
  PSM_DPP_ADD(0x111, 0xf);   // row_shr:1
  PSM_DPP_ADD(0x112, 0xf);   // row_shr:2
  PSM_DPP_ADD(0x114, 0xf);   // row_shr:4
  PSM_DPP_ADD(0x118, 0xf);   // row_shr:8  -> lane 15 of each row of 16 holds the row total
  PSM_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  PSM_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
#undef PSM_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int C_OUT>
__global__ __launch_bounds__(256) void psm_strips_kernel(PsmStripArgs a) {
  constexpr int RB = PSM_STRIP_BAND;                 // rows per band
  constexpr int RPT = RB / 2;                        // rows per thread (two half-bands of 128 columns)
  __shared__ float2 colred[2][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, band = blockIdx.y, cs = blockIdx.z;
  const int S = a.S, SS = S * S;
  const int c = tid & 127, half = tid >> 7;
  const int rbase = band * RB + half * RPT;
  const float* self = a.pred + ((int64_t)(cs * a.B + b) * SS + (int64_t)rbase * S + c) * C_OUT;
  const float* prev = b > 0 ? self - (int64_t)SS * C_OUT : self;
  const float* gm = a.grid + (((int64_t)cs * a.Ny + a.blk_y0x0[2 * b] + rbase) * a.Nx + a.blk_y0x0[2 * b + 1] + c) * a.c_in + a.sdf_ch;
  PSM_STAMP(0, 28);
  __shared__ int32_t tab[C_NS * 6];                 // this block's strip rectangles
  float vs[RPT][C_OUT], vp[RPT][C_OUT];
  bool on[RPT];
  // decoded values and the rectangle table first: their addresses need nothing but the launch
  // arguments; the mask loads wait for the block's grid origin (a dependent scalar load)
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
#pragma unroll
    for (int f = 0; f < C_OUT; ++f) {
      vs[k][f] = self[(int64_t)k * S * C_OUT + f];
      vp[k][f] = prev[(int64_t)k * S * C_OUT + f];
    }
  }
  const int32_t tabv = a.strips[(int64_t)b * a.NS * 6 + min(tid, a.NS * 6 - 1)];
  __builtin_amdgcn_sched_barrier(0);
  float gv[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) gv[k] = gm[(int64_t)k * a.Nx * a.c_in];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < RPT; ++k) on[k] = gv[k] != 0.f;
  if (tid < a.NS * 6) tab[tid] = tabv;
  const int NS = a.NS;
  // per-thread totals over its RPT rows (flow cells only / all cells), then per-COLUMN totals of
  // the band in LDS: a rectangle covering the band's rows completely (the usual case) is then a
  // sum of column totals over [c0, c1), done by ONE wave per slot
  constexpr int NQ = 3 * C_OUT + 1;                  // masked self [C], masked prev [C], unmasked self [C], flow-cell count
  __shared__ float colT[NQ][2][128];
  __shared__ float fin[C_NS][3];
  __shared__ float wsum[4][C_NS][3];
  float tot_s[C_OUT], tot_p[C_OUT], tot_cnt = 0.f, all_s[C_OUT];
#pragma unroll
  for (int f = 0; f < C_OUT; ++f) { tot_s[f] = 0.f; tot_p[f] = 0.f; all_s[f] = 0.f; }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
#pragma unroll
    for (int f = 0; f < C_OUT; ++f) {
      tot_s[f] += on[k] ? vs[k][f] : 0.f;
      tot_p[f] += on[k] ? vp[k][f] : 0.f;
      all_s[f] += vs[k][f];
    }
    tot_cnt += on[k] ? 1.f : 0.f;
  }
#pragma unroll
  for (int f = 0; f < C_OUT; ++f) {
    colT[f][half][c] = tot_s[f];
    colT[C_OUT + f][half][c] = tot_p[f];
    colT[2 * C_OUT + f][half][c] = all_s[f];
  }
  colT[3 * C_OUT][half][c] = tot_cnt;
  __syncthreads();
  PSM_STAMP(0, 29);                                  // loads landed, column totals in LDS
  const int32_t* st = tab;
  float4* outp = a.spart + (((int64_t)cs * a.B + b) * a.n_bands + band) * NS;
  static_assert(C_NS <= 12, "three slots per wave");
  // slots whose rectangle cuts this band (row tests needed): found once, by every wave
  unsigned long long pmask;
  {
    const int sl = min(lane, NS - 1);
    const int r0 = st[6 * sl + 2], r1 = st[6 * sl + 3], c0 = st[6 * sl + 4], c1 = st[6 * sl + 5];
    const bool live = !(r1 <= band * RB || r0 >= (band + 1) * RB || c1 <= c0);
    const bool whole = (r0 <= band * RB && r1 >= (band + 1) * RB);
    pmask = __ballot(lane < NS && live && !whole);
  }
  // every other slot is a sum of column totals over [c0, c1): wave w takes slots w, w+4, w+8 --
  // straight-line (selects, no branches), nine wave sums interleaved
  float rs[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = wave + 4 * j, sv = min(s, NS - 1);
    const int data = st[6 * sv], mask = st[6 * sv + 1], r0 = st[6 * sv + 2], r1 = st[6 * sv + 3], c0 = st[6 * sv + 4], c1 = st[6 * sv + 5];
    const bool live = !(r1 <= band * RB || r0 >= (band + 1) * RB || c1 <= c0);
    const bool whole = (r0 <= band * RB && r1 >= (band + 1) * RB);
    const bool ok = (s < NS) && live && whole;
    const int qb = mask < 0 ? 2 * C_OUT : (data != b ? C_OUT : 0);
    float s0 = 0.f, s1 = 0.f, cnt = 0.f;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int cc = lane + 64 * h2;
      const bool in = ok && (cc >= c0 && cc < c1);
      const float t0 = colT[qb][0][cc] + colT[qb][1][cc];
      const float t1 = colT[qb + C_OUT - 1][0][cc] + colT[qb + C_OUT - 1][1][cc];
      const float tc = colT[3 * C_OUT][0][cc] + colT[3 * C_OUT][1][cc];
      s0 += in ? t0 : 0.f;
      s1 += in ? t1 : 0.f;
      cnt += in ? (mask < 0 ? (float)RB : tc) : 0.f;
    }
    rs[j][0] = s0; rs[j][1] = s1; rs[j][2] = cnt;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    rs[j][0] = wave_sum(rs[j][0]);
    if (C_OUT > 1) rs[j][1] = wave_sum(rs[j][1]); else rs[j][1] = rs[j][0];
    rs[j][2] = wave_sum(rs[j][2]);
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int s = wave + 4 * j;
      if (s < C_NS) { fin[s][0] = rs[j][0]; fin[s][1] = rs[j][1]; fin[s][2] = rs[j][2]; }
    }
  }
  for (unsigned long long pm = pmask; pm; pm &= pm - 1) {     // band cut by the rectangle: row tests, whole workgroup
    const int s = __ffsll((long long)pm) - 1;
    const int data = st[6 * s], mask = st[6 * s + 1], r0 = st[6 * s + 2], r1 = st[6 * s + 3], c0 = st[6 * s + 4], c1 = st[6 * s + 5];
    const bool use_prev = (data != b);
    float s0 = 0.f, s1 = 0.f, cnt = 0.f;
    if (c >= c0 && c < c1) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = rbase + k;
        if (r >= r0 && r < r1 && (mask < 0 || on[k])) {
          s0 += use_prev ? vp[k][0] : vs[k][0];
          if (C_OUT > 1) s1 += use_prev ? vp[k][C_OUT - 1] : vs[k][C_OUT - 1];
          cnt += 1.f;
        }
      }
    }
    s0 = wave_sum(s0); if (C_OUT > 1) s1 = wave_sum(s1); cnt = wave_sum(cnt);
    if (lane == 0) { wsum[wave][s][0] = s0; wsum[wave][s][1] = s1; wsum[wave][s][2] = cnt; }
  }
  __syncthreads();
  PSM_STAMP(0, 30);
  if (tid < NS) {
    const int s = tid;
    if ((pmask >> s) & 1ull)
      outp[s] = make_float4((wsum[0][s][0] + wsum[1][s][0]) + (wsum[2][s][0] + wsum[3][s][0]),
                            (wsum[0][s][1] + wsum[1][s][1]) + (wsum[2][s][1] + wsum[3][s][1]),
                            (wsum[0][s][2] + wsum[1][s][2]) + (wsum[2][s][2] + wsum[3][s][2]), 0.f);
    else
      outp[s] = make_float4(fin[s][0], fin[s][1], fin[s][2], 0.f);
  }
  // gradp: per-column sums of block 0, field 0 (first column holding a flow cell, UGP:294-300)
  if (a.colpart && b == 0) {
    float s0 = 0.f, cnt = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      if (on[k]) { s0 += vs[k][0]; cnt += 1.f; }
    colred[half][c] = make_float2(s0, cnt);
    __syncthreads();
    if (half == 0) {
      const float2 u = colred[0][c], v = colred[1][c];
      a.colpart[((int64_t)cs * a.n_bands + band) * 128 + c] = make_float2(u.x + v.x, u.y + v.y);
    }
  }
}

hipError_t psm_launch_strips(const PsmStripArgs& a, int n_cases, hipStream_t st) {
  if (a.S != 128) return hipErrorInvalidValue;
  if (a.c_out == 1) PSM_LAUNCH((psm_strips_kernel<1>), dim3(a.B, a.n_bands, n_cases), dim3(256), 0, st, a);
  else PSM_LAUNCH((psm_strips_kernel<2>), dim3(a.B, a.n_bands, n_cases), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// chain (+ global shift): one workgroup per case.  All strip partials are combined into
// LDS by the whole workgroup, lane 0 of wave f runs the serial recurrence of field f.
// ---------------------------------------------------------------------------
// Row-parallel form of the offset chain (one wave per field, lane = position of the block
// in its row).  Blocks of one row only depend on each other through the value handed from the
// previously enumerated block (c_prev in SMD/UGP, BC_ant_0 / BC_alter in PM); inside a row
// that hand-over is needed by the whole first row and, elsewhere, only by blocks whose
// BC_ups entry is NaN.  So every row is evaluated lane-parallel from the block above
// (BC_ups is lane-local), followed by an in-order fix-up loop over just the lanes that need
// the hand-over -- the same arithmetic, in the same order, as the serial recurrence
// (psm_chain_v in psm_plan.h, which stays the host replay and the fallback for > 64 columns).
template <int VARIANT>
__device__ __forceinline__ void psm_chain_rows(const PsmChainParams& P, const float* smean, const float* scnt,
                                               const PsmBlock* blk, int field, int lane, float* offs_out) {
  const int n_x = P.n_x, n_y = P.n_y;
  const int ncol = (VARIANT == PSMV_CHAPTER5) ? n_x + 2 : n_x + 1;
  const int nrow = n_y + 2;
  const bool act = lane < ncol;
  const int l = act ? lane : 0;
  int tj;
  if (VARIANT == PSMV_GRADP) tj = l;
  else if (VARIANT == PSMV_DELTAS) tj = n_x - l;
  else tj = (l <= n_x) ? n_x - l : -1;
  const float ref = P.ref_bc;
  float first_col = NAN;                                      // UGP:294-300
  if (VARIANT == PSMV_GRADP && field == 0)
    for (int c = 0; c < 128; ++c)
      if (scnt[P.col_base + c] > 0.f) { first_col = smean[P.col_base + c]; break; }
  float up = (VARIANT == PSMV_CHAPTER5 && tj == -1) ? NAN : 0.f;   // BC_ups[tj] / BC_up_
  float carry = (VARIANT == PSMV_CHAPTER5) ? NAN : 0.f;             // c_prev | BC_ant_0 / BC_alter
  auto rl = [](float v, int q) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), q)); };
  // The strip means of a row's blocks do not depend on the chain: all C_NS of them (plus one
  // count) are read a row ahead, unconditionally, so that the recurrence itself runs on registers.
  constexpr int NSV = (VARIANT == PSMV_DELTAS) ? (int)D_NS : (VARIANT == PSMV_GRADP ? (int)G_NS : (int)C_NS);   // == P.NS
  const int bmax = nrow * ncol - 1;
  auto fetch = [&](float (&M)[NSV], float& cnt_up, int r) {
    const float* mp = smean + min(r * ncol + l, bmax) * NSV;
#pragma unroll
    for (int s = 0; s < NSV; ++s) M[s] = mp[s];
    cnt_up = (VARIANT == PSMV_DELTAS) ? scnt[min(r * ncol + l, bmax) * NSV + D_ROWS_UP] : 0.f;
  };
  float M[NSV], cnt_up;
  fetch(M, cnt_up, 0);
  for (int r = 0; r < nrow; ++r) {
    float Mn[NSV], cnt_up_n;
    fetch(Mn, cnt_up_n, r + 1);                               // clamped on the last row
    const int b = r * ncol + l;
    const bool first = (r == 0), last = (r == n_y + 1);
    if (last && P.skip_last) {                                // duplicate last row left out (uniform)
      if (act) offs_out[b] = NAN;
      continue;
    }
    const float* mp = M;
    const bool unan = (up != up);
    float A, Bm = 0.f, L = 0.f, c;
    bool need;
    if (VARIANT == PSMV_DELTAS) {
      const bool lim = (tj == 0);
      A = lim ? mp[D_CUR_R_LIM] : mp[D_CUR_R_OV];
      Bm = lim ? mp[D_PREV_L_LIM] : mp[D_PREV_L_OV];
      if (first) { c = mp[D_COL_LAST] - ref; need = (lane != 0); }
      else if (!last) { c = mp[D_TOP] - up; need = unan && !(tj != 0 && tj == n_x); }
      else {
        const bool use_side = cnt_up / 16384.f > 0.9f;         // SMD:307
        c = (tj == n_x) ? mp[D_ROWS_UP] - up : mp[D_ROWS_HEAD] - up;
        need = (tj != n_x) && use_side;
      }
    } else if (VARIANT == PSMV_GRADP) {
      const bool lim = (tj == n_x);
      A = lim ? mp[G_CUR_L_LIM] : mp[G_CUR_L_OV];
      Bm = lim ? mp[G_PREV_R_LIM] : mp[G_PREV_R_OV];
      if (first) { c = (field == 0 ? first_col : mp[G_ROW1]) - ref; need = (lane != 0); }
      else { c = (last ? mp[G_ROWS_UP] : mp[G_TOP]) - up; need = unan; }
    } else {
      const bool m1 = (tj == -1), nx = (tj == n_x);
      A = (first && m1) ? mp[C_COLS_C] : mp[C_COLS_R];
      L = mp[C_COLS_0];
      if (first) { c = mp[C_COLS_R] - 0.f; need = !nx; }
      else if (!last) { c = (m1 ? mp[C_TOPC] : mp[C_TOP]) - up; need = !m1 && unan; }
      else { c = (m1 ? mp[C_TC] : mp[C_ROWS_T]) - up; need = !m1 && unan; }
    }
    unsigned long long todo = __ballot(need && act);
    while (todo) {                                            // in enumeration order
      const int q = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const float out_prev = (VARIANT == PSMV_CHAPTER5) ? L - c : c;
      const float cin = (q == 0) ? carry : rl(out_prev, q > 0 ? q - 1 : 0);
      const float cnew = (VARIANT == PSMV_CHAPTER5) ? A - cin : A - (Bm - cin);
      c = (lane == q) ? cnew : c;
    }
    carry = rl((VARIANT == PSMV_CHAPTER5) ? L - c : c, ncol - 1);
    if (VARIANT == PSMV_DELTAS) {
      if (!last) up = ((!first && r == n_y) ? mp[D_ROWS_PI] : mp[D_BOTTOM]) - c;
    } else if (VARIANT == PSMV_GRADP) {
      if (!last) up = ((!first && r == n_y) ? mp[G_ROWS_PI] : mp[G_BOTTOM]) - c;
    } else {
      const bool m1 = (tj == -1), nx = (tj == n_x);
      if (first) up = (nx ? mp[C_RR] : (m1 ? mp[C_RC] : mp[C_ROWS_R])) - c;
      else if (!last) up = (m1 ? mp[C_RC_UNMASKED] : mp[C_ROWS_R]) - c;
    }
    if (act) offs_out[b] = c;
#pragma unroll
    for (int s = 0; s < NSV; ++s) M[s] = Mn[s];
    cnt_up = cnt_up_n;
  }
}

__device__ __forceinline__ void psm_chain_wave(const PsmChainParams& P, const float* smean, const float* scnt,
                                               const PsmBlock* blk, int field, int lane, float* offs_out) {
  if (P.variant == PSMV_DELTAS) psm_chain_rows<PSMV_DELTAS>(P, smean, scnt, blk, field, lane, offs_out);
  else if (P.variant == PSMV_GRADP) psm_chain_rows<PSMV_GRADP>(P, smean, scnt, blk, field, lane, offs_out);
  else psm_chain_rows<PSMV_CHAPTER5>(P, smean, scnt, blk, field, lane, offs_out);
}

__global__ __launch_bounds__(512) void psm_chain_kernel(PsmChainArgs a) {
  constexpr int NB = 128 / PSM_STRIP_BAND;            // bands per block
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cs = blockIdx.x;
  const int B = a.cp.B, SS = a.cp.S * a.cp.S, NS = a.cp.NS, C = a.c_out;
  const int nst = a.n_strips;                         // B*NS (+128 column strips for gradp)
  float* smean = sm;                                  // [C][nst]  sum/count (0/0 -> NaN like np.mean([]))
  float* scnt = smean + C * nst;                      // [nst]
  float* offs = scnt + nst;                           // [C][B]
  float* up = offs + C * B;                           // [C][PSM_MAX_COLS] (fallback path only)
  float* wred = up + C * PSM_MAX_COLS;                // [8]
  const float4* sp = a.spart + (int64_t)cs * B * NB * NS;
  for (int idx = tid; idx < B * NS; idx += 512) {
    const int b = idx / NS, s = idx - b * NS;
    float4 v[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) v[q] = sp[((int64_t)b * NB + q) * NS + s];   // all bands in flight
    float s0 = 0.f, s1 = 0.f, cn = 0.f;
#pragma unroll
    for (int q = 0; q < NB; ++q) { s0 += v[q].x; s1 += v[q].y; cn += v[q].z; }
    smean[idx] = s0 / cn;
    if (C > 1) smean[nst + idx] = s1 / cn;
    scnt[idx] = cn;
  }
  if (a.colpart) {
    for (int c = tid; c < 128; c += 512) {
      float2 v[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) v[q] = a.colpart[((int64_t)cs * NB + q) * 128 + c];
      float s0 = 0.f, cn = 0.f;
#pragma unroll
      for (int q = 0; q < NB; ++q) { s0 += v[q].x; cn += v[q].y; }
      smean[B * NS + c] = s0 / cn;
      if (C > 1) smean[nst + B * NS + c] = 0.f;
      scnt[B * NS + c] = cn;
    }
  }
  for (int idx = tid; idx < C * PSM_MAX_COLS; idx += 512) up[idx] = 0.f;
  __syncthreads();
  if (wave < C) {
    if (a.cp.n_x + 2 <= 64) {
      psm_chain_wave(a.cp, smean + wave * nst, scnt, a.blocks, wave, lane, offs + wave * B);
    } else if (lane == 0) {
      PsmArrayChainCtx<float> cx{a.blocks, smean + wave * nst, scnt, NS, a.cp.col_base, a.cp.S, up + wave * PSM_MAX_COLS, offs + wave * B};
      psm_chain<float>(a.cp, cx, wave);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < C * B; idx += 512) a.offs[(int64_t)cs * C * B + idx] = offs[idx];
  for (int f = 0; f < C; ++f) {
    const int L = a.shiftL[f];
    const int32_t* la = a.shiftA + (int64_t)f * a.Lmax;
    const int32_t* lb = a.shiftB + (int64_t)f * a.Lmax;
    const float* pred = a.pred + ((int64_t)cs * B * SS) * C + f;
    float acc = 0.f;
    for (int k = tid; k < L; k += 512) {
      const int oa = a.owner[la[k]], ob = a.owner[lb[k]];
      const float va = oa >= 0 ? pred[(int64_t)oa * C] - offs[f * B + oa / SS] : 0.f;
      const float vb = ob >= 0 ? pred[(int64_t)ob * C] - offs[f * B + ob / SS] : 0.f;
      acc += 3.f * va - vb;
    }
    acc = wave_sum(acc);
    __syncthreads();
    if (lane == 0) wred[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < 8; ++w) t += wred[w];
      a.shift[cs * C + f] = t / (float)L / 3.f;
    }
  }
}

hipError_t psm_launch_chain(const PsmChainArgs& a, int n_cases, hipStream_t st) {
  const size_t lds = ((size_t)a.c_out * a.n_strips + a.n_strips + (size_t)a.c_out * a.cp.B + (size_t)a.c_out * PSM_MAX_COLS + 8) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  PSM_LAUNCH(psm_chain_kernel, dim3(n_cases), dim3(512), lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// assemble = chain + shift + paste in one launch (small block counts): every paste
// workgroup re-runs the (cheap, register-resident) offset chain instead of waiting for a
// separate one-workgroup launch.  The global shift is split into a part that does not
// depend on the offsets (gathered while the strip partials are in flight) and a weighted
// sum of the offsets:  shift = sum_k(3 pred[A_k] - pred[B_k])/(3L) - sum_b w_b offs_b.
// ---------------------------------------------------------------------------
// Workgroup barrier that only drains LDS traffic: __syncthreads() also waits for every outstanding
// global load (vmcnt(0)), which would serialise the gathers below with the chain.
#define PSM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int C>
__global__ __launch_bounds__(256) void psm_assemble_kernel(PsmChainArgs a, PsmPasteArgs p) {
  constexpr int NB = 128 / PSM_STRIP_BAND;
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cs = blockIdx.y;
  const int B = a.cp.B, SS = a.cp.S * a.cp.S, NS = a.cp.NS;
  const int nst = a.n_strips;
  float* smean = sm;                                  // [C][nst]
  float* scnt = smean + C * nst;                      // [nst]
  float* offs = scnt + nst;                           // [C][B]
  float* wred = offs + C * B;                         // [C][4] + [C] shift
  PSM_STAMP(0, 36);
  // ---- phase 1a: every independent load (cell owner, shift-list owners, strip partials).
  // Straight-line: indices are clamped and results selected, so that all loads of a phase are
  // in flight together (a conditional load costs a branch and, with it, a drained vmcnt).
  const int pix = blockIdx.x * 256 + tid;
  const int o_raw = a.owner[min(pix, p.npix - 1)];
  int la[C], lb[C];                                  // one shift-list entry per thread and field (L <= 256 on this path)
#pragma unroll
  for (int f = 0; f < C; ++f) {
    const int kk = (int)((int64_t)f * a.Lmax) + min(tid, max(a.shiftL[f], 1) - 1);
    la[f] = a.shiftOwnA[kk];
    lb[f] = a.shiftOwnB[kk];
  }
  const float4* sp = a.spart + (int64_t)cs * B * NB * NS;
  const int idx0 = tid, idx1 = tid + 256;             // B*NS <= 64*11 = 704 -> at most 3 per thread
  v4f pv0[NB], pv1[NB];
  {
    const int b0 = min(idx0, B * NS - 1) / NS, s0 = min(idx0, B * NS - 1) - b0 * NS;
    const int b1 = min(idx1, B * NS - 1) / NS, s1 = min(idx1, B * NS - 1) - b1 * NS;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      pv0[q] = *reinterpret_cast<const v4f*>(sp + ((int64_t)b0 * NB + q) * NS + s0);
      pv1[q] = *reinterpret_cast<const v4f*>(sp + ((int64_t)b1 * NB + q) * NS + s1);
    }
  }
  float2 cp[NB];
  if (a.colpart) {                                    // uniform
#pragma unroll
    for (int q = 0; q < NB; ++q) cp[q] = a.colpart[((int64_t)cs * NB + q) * 128 + (tid & 127)];
  }
  const float w_shift = a.shiftW[min(wave, C - 1) * B + min(lane, B - 1)];   // used after the chain (waves < C)
  __builtin_amdgcn_sched_barrier(0);
  // ---- phase 1b: dependent gathers from the decoded blocks.  They are NOT waited for before the
  // chain: the barriers below are LDS-only (s_waitcnt lgkmcnt(0); s_barrier), so these loads land
  // while the offset chain runs.
  const int o = pix < p.npix ? o_raw : -1;
  const float* predc = a.pred + ((int64_t)cs * B * SS) * C;
  float src[C], ga[C], gb[C];
#pragma unroll
  for (int f = 0; f < C; ++f) {
    const bool in = tid < a.shiftL[f];
    la[f] = in ? la[f] : -1;
    lb[f] = in ? lb[f] : -1;
    src[f] = predc[(int64_t)max(o, 0) * C + f];
    ga[f] = predc[(int64_t)max(la[f], 0) * C + f];
    gb[f] = predc[(int64_t)max(lb[f], 0) * C + f];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- strip partials -> means
  auto fold = [&](const v4f (&v)[NB], int idx) {
    float s0 = 0.f, s1 = 0.f, cn = 0.f;
#pragma unroll
    for (int q = 0; q < NB; ++q) { s0 += v[q].x; s1 += v[q].y; cn += v[q].z; }
    const float m0 = s0 / cn, m1 = s1 / cn;
    if (idx < B * NS) {
      smean[idx] = m0;
      if (C > 1) smean[nst + idx] = m1;
      scnt[idx] = cn;
    }
  };
  fold(pv0, idx0);
  fold(pv1, idx1);
  for (int idx = tid + 512; idx < B * NS; idx += 256) {   // only for > 46 blocks
    const int b = idx / NS, s = idx - b * NS;
    v4f v[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) v[q] = *reinterpret_cast<const v4f*>(sp + ((int64_t)b * NB + q) * NS + s);
    fold(v, idx);
  }
  if (a.colpart) {
    float s0 = 0.f, cn = 0.f;
#pragma unroll
    for (int q = 0; q < NB; ++q) { s0 += cp[q].x; cn += cp[q].y; }
    const float m0 = s0 / cn;
    if (tid < 128) {
      smean[B * NS + tid] = m0;
      if (C > 1) smean[nst + B * NS + tid] = 0.f;
      scnt[B * NS + tid] = cn;
    }
  }
  PSM_LDS_BARRIER();
  PSM_STAMP(0, 37);
  float t_shift = 0.f;
  if (wave < C) {
    PSM_STAMP(0, 38);
    psm_chain_wave(a.cp, smean + wave * nst, scnt, a.blocks, wave, lane, offs + wave * B);
    PSM_STAMP(0, 39);
    // shift of this field: weighted sum of the offsets (B <= 64 on this path); same-wave LDS
    // writes above are visible to the wave's own later reads
    const float t = (lane < B && w_shift != 0.f) ? w_shift * offs[wave * B + lane] : 0.f;
    t_shift = wave_sum(t);
  }
  // offset-independent part of the shift (the gathers have landed under the chain)
  float pp[C];
#pragma unroll
  for (int f = 0; f < C; ++f) {
    float acc = 3.f * (la[f] >= 0 ? ga[f] : 0.f) - (lb[f] >= 0 ? gb[f] : 0.f);
    for (int k = tid + 256; k < a.shiftL[f]; k += 256) {   // lists longer than 256 (not on the small-grid path)
      const int ia = a.shiftOwnA[(int64_t)f * a.Lmax + k], ib = a.shiftOwnB[(int64_t)f * a.Lmax + k];
      acc += 3.f * (ia >= 0 ? predc[(int64_t)ia * C + f] : 0.f) - (ib >= 0 ? predc[(int64_t)ib * C + f] : 0.f);
    }
    pp[f] = wave_sum(acc);
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < C; ++f) wred[f * 4 + wave] = pp[f];
  }
  PSM_LDS_BARRIER();
  if (wave < C && lane == 0) {
    const float part = (wred[wave * 4 + 0] + wred[wave * 4 + 1]) + (wred[wave * 4 + 2] + wred[wave * 4 + 3]);
    wred[4 * C + wave] = part / (float)a.shiftL[wave] / 3.f - t_shift;
  }
  PSM_LDS_BARRIER();
  PSM_STAMP(0, 40);
  if (blockIdx.x == 0) {       // introspection copies (psm_read_stage)
    for (int idx = tid; idx < C * B; idx += 256) a.offs[(int64_t)cs * C * B + idx] = offs[idx];
    if (tid < C) a.shift[cs * C + tid] = wred[4 * C + tid];
  }
  if (pix >= p.npix) return;
  float* out = p.fields + ((int64_t)cs * p.npix + pix) * C;
  if (o < 0) {
#pragma unroll
    for (int f = 0; f < C; ++f) out[f] = 0.f;
    return;
  }
  const int b = o / SS;
#pragma unroll
  for (int f = 0; f < C; ++f) out[f] = src[f] - offs[f * B + b] - wred[4 * C + f];
  PSM_STAMP(0, 41);
}

hipError_t psm_launch_assemble(const PsmChainArgs& a, const PsmPasteArgs& p, int n_cases, hipStream_t st) {
  const size_t lds = ((size_t)a.c_out * a.n_strips + a.n_strips + (size_t)a.c_out * a.cp.B + 6 * a.c_out + 8) * sizeof(float);
  const dim3 grid((p.npix + 255) / 256, n_cases);
  if (a.c_out == 1) PSM_LAUNCH((psm_assemble_kernel<1>), grid, dim3(256), lds, st, a, p);
  else PSM_LAUNCH((psm_assemble_kernel<2>), grid, dim3(256), lds, st, a, p);
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
  PSM_LAUNCH(psm_paste_kernel, dim3((a.npix + 255) / 256, n_cases), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// geometry-bound fast path (psm_bind_geometry; structs in psm_kernels.h)
// ---------------------------------------------------------------------------
// Table build, once per geometry.  Row layout: [c_out][nst] strips, then [c_out][B] shift rows.
//   strip row (f, s):  G[k] = sum over the rectangle of block `data`, cells that are flow cells of block `mask`,
//                      of comp[k][(r*S + c)*C + f];  M = the same sum of mean;  cnt = number of such cells
//   shift row (f, b):  G[k] = sum_i 3 comp[k][A_i] - sum_i comp[k][B_i] over the shift-list entries owned by block b
// One workgroup per row, thread = component k (double accumulation: this runs once, not per solve).
__global__ __launch_bounds__(128) void psm_bind_rows_kernel(PsmBindArgs a) {
  const int C = a.c_out, S = a.S, SS = S * S, K_out = SS * C;
  const int row = blockIdx.x, n_strip_rows = C * a.nst;
  const int tid = threadIdx.x;
  double acc[4] = {0, 0, 0, 0};                       // components tid, tid+128, ... (ld_out <= 512)
  double msum = 0.0, cnt = 0.0;
  int blk_of;
  if (row < n_strip_rows) {
    const int f = row / a.nst, s = row - f * a.nst;
    const int32_t* st = a.strips + 6 * s;
    const int data = st[0], mask = st[1], r0 = st[2], r1 = st[3], c0 = st[4], c1 = st[5];
    blk_of = data;
    const int my0 = mask >= 0 ? a.blk_y0x0[2 * mask] : 0, mx0 = mask >= 0 ? a.blk_y0x0[2 * mask + 1] : 0;
    for (int r = r0; r < r1; ++r)
      for (int c = c0; c < c1; ++c) {
        const bool on = mask < 0 || a.grid[((int64_t)(my0 + r) * a.Nx + (mx0 + c)) * a.c_in + a.sdf_ch] != 0.f;
        if (!on) continue;                              // uniform
        const int col = (r * S + c) * C + f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (tid + 128 * u < a.ld_out) acc[u] += (double)a.comp[(int64_t)(tid + 128 * u) * K_out + col];
        msum += (double)a.mean[col];
        cnt += 1.0;
      }
  } else {
    const int q = row - n_strip_rows, f = q / a.B, b = q - f * a.B;
    blk_of = b;
    cnt = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int32_t* list = (pass == 0 ? a.shiftOwnA : a.shiftOwnB) + (int64_t)f * a.Lmax;
      const double w = pass == 0 ? 3.0 : -1.0;
      for (int i = 0; i < a.shiftL[f]; ++i) {
        const int o = list[i];
        if (o < 0 || o / SS != b) continue;            // uniform
        const int col = (o - b * SS) * C + f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (tid + 128 * u < a.ld_out) acc[u] += w * (double)a.comp[(int64_t)(tid + 128 * u) * K_out + col];
        msum += w * (double)a.mean[col];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (tid + 128 * u < a.ld_out) a.G[(int64_t)row * a.ld_out + tid + 128 * u] = acc[u];
  if (tid == 0) { a.Mrow[row] = msum; a.cnt[row] = (float)cnt; a.row_of[row] = a.row_base + blk_of; }
}

// fold the head layer into the rows:  g2[row][j] = sum_k Wh[j][k] sa[k] G[row][k];
// c2[row] = sum_k (bh[k] sa[k] + sb[k]) G[row][k] + M[row]   (head: out = (act @ Wh + bh) * sa + sb)
__global__ __launch_bounds__(256) void psm_bind_fold_kernel(PsmBindArgs a) {
  extern __shared__ double gs[];                        // [ld_out] G row scaled by sa
  const int row = blockIdx.x, tid = threadIdx.x;
  double part = 0.0;
  for (int k = tid; k < a.ld_out; k += 256) {
    const double g = a.G[(int64_t)row * a.ld_out + k];
    gs[k] = g * (double)a.sa[k];
    part += ((double)a.bh[k] * (double)a.sa[k] + (double)a.sb[k]) * g;
  }
  __shared__ double red[256];
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  if (tid == 0) a.c2[row] = (float)(red[0] + a.Mrow[row]);
  for (int j = tid; j < a.Kh; j += 256) {
    const float* w = a.Wh + (int64_t)j * a.ldw;
    double acc = 0.0;
    for (int k = 0; k < a.ld_out; ++k) acc += (double)w[k] * gs[k];
    a.g2[(int64_t)row * a.Kh + j] = (float)acc;
  }
}

__global__ __launch_bounds__(256) void psm_bind_own_kernel(PsmBindArgs a) {
  const int SS = a.S * a.S, wpb = SS / 32;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= a.B * wpb) return;
  const int b = w / wpb, p0 = (w - b * wpb) * 32;
  const int y0 = a.blk_y0x0[2 * b], x0 = a.blk_y0x0[2 * b + 1];
  uint32_t bits = 0;
  for (int t = 0; t < 32; ++t) {
    const int px = p0 + t, r = px / a.S, c = px - r * a.S;
    if (a.owner[(int64_t)(y0 + r) * a.Nx + (x0 + c)] == b * SS + px) bits |= 1u << t;
  }
  a.ownbits[w] = bits;
}

hipError_t psm_launch_bind(const PsmBindArgs& a, hipStream_t st) {
  if (a.ld_out > 512 || a.ld_out < 1 || (a.S * a.S) % 32 != 0) return hipErrorInvalidValue;
  const int rows = a.c_out * a.nst + a.c_out * a.B;
  PSM_LAUNCH(psm_bind_rows_kernel, dim3(rows), dim3(128), 0, st, a);
  PSM_LAUNCH(psm_bind_fold_kernel, dim3(rows), dim3(256), (size_t)a.ld_out * sizeof(double), st, a);
  PSM_LAUNCH(psm_bind_own_kernel, dim3((a.B * (a.S * a.S / 32) + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

// decode + offset chain + paste.  Waves 0-3: the decode tile of psm_decode128_kernel (one row chunk); waves 4, 5: the
// offset chain of field 0 / 1 from the strip means the head launch left in `dots`, and the global shift
// (shift_f = sum_b dots_shift[f][b] / (3 L_f) - sum_b w_b offs_b).  Both run while the other's loads are in flight;
// the epilogue writes value - offset - shift for the block pixels that own their cell (ownership bits) straight
// into the field -- the decoded blocks are never stored.
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pk_bf16x4 __attribute__((ext_vector_type(4)));
// BF (bf16 handles): the tile is rounded to bf16 on its way into LDS and multiplied with the bf16 basis by
// v_mfma_f32_32x32x16_bf16, exactly like psm_decode_bf16_kernel (psm_bf16.hip) -- same rounding points.
template <int MTC, int C, int LDR, int MODE>     // LDR = ld_res: output components padded to 32, 64, 96 or 128; MODE 0 f32, 1 bf16 handle, 2 x6
__global__ __launch_bounds__(384) void psm_decode_paste_kernel(PsmDecodeArgs a, PsmBoundArgs p) {
  constexpr bool BF = MODE == 1, X6 = MODE == 2;
  constexpr int LDX = LDR + 4;                         // x6: plane row stride in bf16
  constexpr int LDA = X6 ? 3 * LDX / 2 : (BF ? (LDR + 8) / 2 : LDR + 4);    // tile floats per row (bf16: LDR + 8 halves; x6: three planes)
  constexpr int Q = LDR / 4, GD = BF ? LDR / 16 : LDR / 8, NA = MTC * 32 * Q / 256;
  constexpr int WPB = (128 / C) / 32;                  // ownership words per block for this workgroup's 128 columns
  constexpr int NST = 8;                               // staging rounds of 384 floats (C*nst + nst <= 3072)
  constexpr int R = MTC * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  psm_warm_kernargs<sizeof(PsmDecodeArgs) + sizeof(PsmBoundArgs)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = p.B, nst = p.n_strips, S = p.cp.S;
  // per block row, one 32-byte record for the epilogue: {element offset of the block's first cell (bits), -, out_scale, -, ownership words (WPB <= 4)}
  float* rec = lds + R * LDA;                          // [R][8]
  float* smean = rec + R * 8;                          // [C][nst]
  float* scnt = smean + C * nst;                       // [nst]
  float* offs = scnt + nst;                            // [C][B]
  float* wred = offs + C * B;                          // [4] shift per field
  const bool dec = wave < 4;                           // uniform per wave
  const int i = lane & 31, h = lane >> 5;
  const int ct = min((int)blockIdx.x * 4 + min(wave, 3), a.n_coltiles - 1);
  const bool live = dec && ((int)blockIdx.x * 4 + wave) < a.n_coltiles;
  PSM_STAMP(0, 20);
  // The FIRST HALF of the basis stream is requested before anything else (round 6): its addresses need nothing but the column tile,
  // while the small operands below cost ~250 instructions of address arithmetic and branches before the stream could start.  They
  // return behind that half (the counter is in order), which is still well before the second half has landed.
  float4 b[GD];                                        // bf16: 8 halves per 16-byte piece
  const float4* bp = a.bpack + ((int64_t)ct * GD) * 64 + lane;
#pragma unroll
  for (int g = 0; g < GD / 2; ++g) b[g] = stream_load(bp + g * 64);
  __builtin_amdgcn_sched_barrier(0);
  // ---- every load of the prologue, clamped and unconditional
  const int n_stage = p.cf ? 1 : C * nst + nst;
  float sv[NST];
  if (!p.cf) {                                         // (closed form: nothing to stage -- eight clamped loads and their address chains less)
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int idx = min(tid + 384 * u, n_stage - 1);
      const float* src = idx < C * nst ? p.dots + idx : p.scnt + (idx - C * nst);
      sv[u] = *src;
    }
  } else {
#pragma unroll
    for (int u = 0; u < NST; ++u) sv[u] = 0.f;
  }
  // closed form of the chain: offset + shift of block b = a0 + the long dot the head launch left (one thread per value)
  const int cfi = min(tid, C * B - 1);
  // (the two terms are added where they are written to LDS: added here, the sum waited vmcnt(0) for both -- a full round trip to
  // what the head launch has just written -- BEFORE the basis stream below was requested)
  float cfa = 0.f, cfb = 0.f;
  if (p.cf) { cfa = p.cf_a0[cfi]; cfb = p.cf_dots[cfi]; }
  const int rb = min(tid, B - 1);                      // threads < B: the record of block row tid
  uint32_t ow[WPB];
#pragma unroll
  for (int w = 0; w < WPB; ++w) ow[w] = p.ownbits[(int64_t)rb * (S * S / 32) + (int)blockIdx.x * WPB + w];
  const int y0v = p.blk_y0x0[2 * rb], x0v = p.blk_y0x0[2 * rb + 1];
  const float rs = a.row_scale[min(rb, a.Mpad - 1)];
  const int cf = min(max(wave - 4, 0), C - 1);         // chain waves: their field
  const float w_shift = p.shiftW[cf * B + min(lane, B - 1)];
  const float s_raw = p.cf ? 0.f : p.dots[C * nst + cf * B + min(lane, B - 1)];
  const float gf0 = p.gflags[min(lane, p.n_gwaves - 1)], gf1 = p.gflags[min(lane + 64, p.n_gwaves - 1)];   // guard flags (0 / NaN)
  v4f x[NA];
#pragma unroll
  for (int u = 0; u < NA; ++u) {                       // (the two chain waves load a clamped duplicate: 384 threads, 256 slots)
    const int idx = min(tid, 255) + 256 * u, row = idx / Q, q = idx - row * Q;
    x[u] = *reinterpret_cast<const v4f*>(a.res + (int64_t)min(row, a.Mpad - 1) * LDR + 4 * q);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = GD / 2; g < GD; ++g) b[g] = stream_load(bp + g * 64);
  const int col = ct * 32 + i;
  const float mu = a.mean[col];
  __builtin_amdgcn_sched_barrier(0);
  PSM_STAMP(0, 24);
  // ---- LDS staging; the barrier drains LDS traffic only, so the basis loads above stay in flight behind it
#pragma unroll
  for (int u = 0; u < NST; ++u)
    if (!p.cf && tid + 384 * u < n_stage) smean[tid + 384 * u] = sv[u];       // smean and scnt are contiguous
  if (p.cf && tid < C * B) offs[tid] = cfa + cfb;
  if (tid < B) {
    float* rr = rec + tid * 8;
    rr[0] = __uint_as_float((uint32_t)((y0v * p.Nx + x0v) * C)); rr[1] = 0.f; rr[2] = rs; rr[3] = 0.f;   // element offset of the block's first cell (one case: < 2^31)
#pragma unroll
    for (int w = 0; w < WPB; ++w) rr[4 + w] = __uint_as_float(ow[w]);
  }
  if (tid < 256) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, row = idx / Q, q = idx - row * Q;
      if constexpr (X6) {
        x6_bf16x4 vh, vm, vl;
        psm_split3(x[u], vh, vm, vl);
        __bf16* dst = reinterpret_cast<__bf16*>(lds) + row * LDX + 4 * q;
        *reinterpret_cast<x6_bf16x4*>(dst) = vh;
        *reinterpret_cast<x6_bf16x4*>(dst + R * LDX) = vm;
        *reinterpret_cast<x6_bf16x4*>(dst + 2 * R * LDX) = vl;
      } else if constexpr (BF) {
        pk_bf16x4 v;
        v[0] = (__bf16)x[u][0]; v[1] = (__bf16)x[u][1]; v[2] = (__bf16)x[u][2]; v[3] = (__bf16)x[u][3];
        *reinterpret_cast<pk_bf16x4*>(reinterpret_cast<__bf16*>(&lds[row * LDA]) + 4 * q) = v;
      } else {
        *reinterpret_cast<v4f*>(&lds[row * LDA + 4 * q]) = x[u];
      }
    }
  }
  PSM_STAMP(0, 25);
  PSM_LDS_BARRIER();
  PSM_STAMP(0, 21);
  f32x16 acc[MTC];
  x6_bf16x8 Bh[X6 ? LDR / 16 : 1], Bm[X6 ? LDR / 16 : 1], Bl[X6 ? LDR / 16 : 1];
  if (dec) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      acc[mt] = (f32x16){0};
      if constexpr (X6) {
        const __bf16* arow = reinterpret_cast<const __bf16*>(lds) + (mt * 32 + i) * LDX + 4 * h;
#pragma unroll
        for (int st = 0; st < LDR / 16; ++st) {
          if (mt == 0) {                               // the basis slice is split on the way: each step waits for its two groups only
            x6_bf16x4 h0, m0, l0, h1, m1, l1;
            psm_split3((f32x4){b[2 * st].x, b[2 * st].y, b[2 * st].z, b[2 * st].w}, h0, m0, l0);
            psm_split3((f32x4){b[2 * st + 1].x, b[2 * st + 1].y, b[2 * st + 1].z, b[2 * st + 1].w}, h1, m1, l1);
            Bh[st] = psm_cat4(h0, h1); Bm[st] = psm_cat4(m0, m1); Bl[st] = psm_cat4(l0, l1);
          }
          const x6_bf16x8 ah = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 16 * st + 8));
          const x6_bf16x8 am = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + R * LDX + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + R * LDX + 16 * st + 8));
          const x6_bf16x8 al = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 2 * R * LDX + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 2 * R * LDX + 16 * st + 8));
          acc[mt] = MFMA_X6(am, Bm[st], acc[mt]);
          acc[mt] = MFMA_X6(al, Bh[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bl[st], acc[mt]);
          acc[mt] = MFMA_X6(am, Bh[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bm[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bh[st], acc[mt]);
        }
      } else if constexpr (BF) {
        const __bf16* arow = reinterpret_cast<const __bf16*>(&lds[(mt * 32 + i) * LDA]) + 8 * h;
#pragma unroll
        for (int g = 0; g < GD; ++g) {
          const pk_bf16x8 av = *reinterpret_cast<const pk_bf16x8*>(arow + 16 * g);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(pk_bf16x8, b[g]), acc[mt], 0, 0, 0);
        }
      } else {
        const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h];
        float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
        for (int g = 0; g < GD; ++g) {
          const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < GD ? g + 1 : g));
          acc[mt] = MFMA32(av.x, b[g].x, acc[mt]);
          acc[mt] = MFMA32(av.y, b[g].y, acc[mt]);
          acc[mt] = MFMA32(av.z, b[g].z, acc[mt]);
          acc[mt] = MFMA32(av.w, b[g].w, acc[mt]);
          av = an;
        }
      }
    }
  } else if (wave - 4 < C) {
    const int f = wave - 4;
    const float guard = psm_guard_sum(p.gflags, p.n_gwaves, lane, gf0, gf1);   // NaN when the grid is not the bound geometry
    if (p.cf) {                                // closed form: the staged values already hold offset + shift
      if (lane == 0) wred[f] = guard;
    } else {
      // (the chain shares its SIMD with an MFMA wave and in effect runs after that wave's 2 us of MFMAs -- 3.7 us to the
      // chain's end instead of 1.6 alone; s_setprio(3) here changes nothing: the vector ALU itself is taken)
      psm_chain_wave(p.cp, smean + f * nst, scnt, p.blocks, f, lane, offs + f * B);
      const float t = (lane < B && w_shift != 0.f) ? w_shift * offs[f * B + lane] : 0.f;   // same-wave LDS writes are visible
      const float t_shift = wave_sum(t);
      const float raw = wave_sum(lane < B ? s_raw : 0.f);
      if (lane == 0) wred[f] = raw / (float)p.shiftL[f] / 3.f - t_shift + guard;
    }
    if (blockIdx.x == 0 && f == 0 && lane == 0) {
#if defined(PSM_STAMPS) && !defined(PSM_STAMPS_ENC)
      g_psm_stamps[39] = __builtin_amdgcn_s_memrealtime();
#endif
    }
  }
  PSM_STAMP(0, 22);
  PSM_LDS_BARRIER();
  if (blockIdx.x == 0 && !p.cf) {       // introspection copies (psm_read_stage; under the closed form it runs the chain itself)
    for (int idx = tid; idx < C * B; idx += 384) p.offs[idx] = offs[idx];
    if (tid < C) p.shift[tid] = wred[tid];
  }
  if (!live) return;
  const float mu_r = psm_settled(mu);
  const int px = col / C, f = col - px * C;
  const int pxl = px - (int)blockIdx.x * (128 / C);
  const int r = px / S, c = px - r * S;
  const float sh = wred[f];
  const uint32_t pix_off = (uint32_t)((r * p.Nx + c) * C + f);       // this lane's cell within a block's window, in elements
  // epilogue in two passes so that the LDS reads of all 16 rows are in flight together: records and offsets first
  // (straight-line), then the owned values are stored
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt) {
    v4f ra[16], rb4[16];
    float of[16];
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int mc = min(mt * 32 + acc_row(rg, h), B - 1);
      ra[rg] = *reinterpret_cast<const v4f*>(rec + mc * 8);
      rb4[rg] = *reinterpret_cast<const v4f*>(rec + mc * 8 + 4);
      of[rg] = offs[f * B + mc];
    }
    auto paste = [&](auto buffered) {
      const __amdgpu_buffer_rsrc_t frs = psm_store_rsrc(p.fields, p.field_bytes);
#pragma unroll
      for (int rg = 0; rg < 16; ++rg) {
        const int m = mt * 32 + acc_row(rg, h);
        const uint32_t word = __float_as_uint(rb4[rg][pxl >> 5]);
        const bool mine = m < B && ((word >> (pxl & 31)) & 1u);
        const float val = (acc[mt][rg] + mu_r) * ra[rg][2] - of[rg] - sh;
        if constexpr (decltype(buffered)::value) psm_store_if(frs, __float_as_uint(ra[rg][0]) + pix_off, mine, val);
        else if (mine) p.fields[(size_t)(__float_as_uint(ra[rg][0]) + pix_off)] = val;
      }
    };
    if (p.field_bytes) paste(std::true_type{}); else paste(std::false_type{});        // uniform
  }
  PSM_STAMP(0, 23);
}

hipError_t psm_launch_decode_paste(const PsmDecodeArgs& a, const PsmBoundArgs& p, int c_out, hipStream_t st, int bf16) {
  if (a.ld_res > 128 || a.ld_res % 32 != 0 || a.Mpad > 64 || a.Mpad % 32 != 0 || p.B > 64 || p.B < 1 || a.M != p.B) return hipErrorInvalidValue;
  if ((c_out != 1 && c_out != 2) || c_out * p.n_strips + p.n_strips > 8 * 384) return hipErrorInvalidValue;
  const int nwg = (a.n_coltiles + 3) / 4, mtc = a.Mpad / 32, wpb = (128 / c_out) / 32;
  (void)wpb;
  const bool x6 = a.x6 && !bf16;
  const size_t tile_floats = x6 ? (size_t)mtc * 32 * (a.ld_res + 4) * 3 / 2 : (size_t)mtc * 32 * (a.ld_res + 4);
  const size_t lds = (tile_floats + (size_t)mtc * 32 * 8 + (size_t)c_out * p.n_strips + p.n_strips + (size_t)c_out * p.B + 4) * sizeof(float);
#define DP(M_, C_, L_)                                                                                              \
  do {                                                                                                              \
    if (bf16) PSM_LAUNCH((psm_decode_paste_kernel<M_, C_, L_, 1>), dim3(nwg), dim3(384), lds, st, a, p);    \
    else if (x6) PSM_LAUNCH((psm_decode_paste_kernel<M_, C_, L_, 2>), dim3(nwg), dim3(384), lds, st, a, p); \
    else PSM_LAUNCH((psm_decode_paste_kernel<M_, C_, L_, 0>), dim3(nwg), dim3(384), lds, st, a, p);         \
  } while (0)
#define DPL(L_)                                                        \
  do {                                                                 \
    if (mtc == 1) { if (c_out == 1) DP(1, 1, L_); else DP(1, 2, L_); } \
    else { if (c_out == 1) DP(2, 1, L_); else DP(2, 2, L_); }          \
  } while (0)
  if (a.ld_res == 32) DPL(32); else if (a.ld_res == 64) DPL(64); else if (a.ld_res == 96) DPL(96); else DPL(128);
#undef DPL
#undef DP
  return hipGetLastError();
}

// ---- case batches on a bound geometry: the chain of every case in one small launch (one workgroup per case, wave f =
// field f), then decode + paste for all block rows (row chunks like psm_decode128_kernel)
template <int C>
__global__ __launch_bounds__(256) void psm_chain_dots_kernel(PsmBoundBatchArgs p) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cs = blockIdx.x;
  const int B = p.B, nst = p.n_strips;
  float* smean = sm;                                   // [C][nst]
  float* scnt = smean + C * nst;                       // [nst]
  float* offs = scnt + nst;                            // [C][B]
  const float* dots = p.dots + (int64_t)cs * p.rows_pc;
  const float* cnt = p.scnt + (int64_t)cs * p.rows_pc;
  const int n_stage = C * nst + nst;
  const float gpart = psm_guard_part(p.gflags, p.n_gwaves, lane, 0);   // guard flags of this solve (0 / NaN): in flight with the staging loads
  for (int base = 0; base < n_stage; base += 256 * 8) {            // 8 loads in flight per thread and round
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = min(base + tid + 256 * u, n_stage - 1);
      sv[u] = idx < C * nst ? dots[idx] : cnt[idx - C * nst];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + tid + 256 * u < n_stage) smean[base + tid + 256 * u] = sv[u];
  }
  __syncthreads();
  if (wave < C) {
    psm_chain_wave(p.cp, smean + wave * nst, scnt, p.blocks, wave, lane, offs + wave * B);
    float t = 0.f, raw = 0.f;
    for (int b = lane; b < B; b += 64) {                            // same-wave LDS writes above are visible
      const float w = p.shiftW[wave * B + b];
      t += w != 0.f ? w * offs[wave * B + b] : 0.f;                   // skipped blocks have NaN offsets and weight 0
      raw += dots[C * nst + wave * B + b];
      p.offs[((int64_t)cs * C + wave) * B + b] = offs[wave * B + b];
    }
    const float t_shift = wave_sum(t), raw_all = wave_sum(raw);
    const float guard = wave_sum(gpart);                              // NaN when a grid of this batch is not its bound geometry
    if (lane == 0) p.shift[cs * C + wave] = raw_all / (float)p.shiftL[wave] / 3.f - t_shift + guard;
  }
}

hipError_t psm_launch_chain_dots(const PsmBoundBatchArgs& p, int c_out, hipStream_t st) {
  const size_t lds = ((size_t)c_out * p.n_strips + p.n_strips + (size_t)c_out * p.B) * sizeof(float);
  if ((c_out != 1 && c_out != 2) || lds > 60 * 1024 || p.B < 1) return hipErrorInvalidValue;
  if (c_out == 1) PSM_LAUNCH((psm_chain_dots_kernel<1>), dim3(p.n_cases), dim3(256), lds, st, p);
  else PSM_LAUNCH((psm_chain_dots_kernel<2>), dim3(p.n_cases), dim3(256), lds, st, p);
  return hipGetLastError();
}

template <int MTC, int C, int LDR, int MODE>           // MODE 0: exact-f32 MFMA, 1: bf16 handle (operands rounded), 2: x6 (float32 accuracy on the bf16 pipe)
__global__ __launch_bounds__(256) void psm_decode_paste_batch_kernel(PsmDecodeArgs a, PsmBoundBatchArgs p, int m_end) {
  psm_warm_kernargs<sizeof(PsmDecodeArgs) + sizeof(PsmBoundBatchArgs)>();
#if defined(PSM_STAMPS) && !defined(PSM_STAMPS_ENC)
#define DSTAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (k) < 40) g_psm_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif
  DSTAMP(0);
  constexpr bool BF = MODE == 1, X6 = MODE == 2;
  constexpr int LDX = LDR + 4;                         // x6: plane row stride in bf16 (LDR / 2 + 2 dwords = 2 * odd: ds_read_b64 conflict-free)
  constexpr int LDA = X6 ? 3 * LDX / 2 : (BF ? (LDR + 8) / 2 : LDR + 4);    // tile floats per row (bf16: LDR + 8 halves; x6: three planes)
  constexpr int Q = LDR / 4, GD = BF ? LDR / 16 : LDR / 8, NA = MTC * 32 * Q / 256;
  constexpr int WPB = (128 / C) / 32, R = MTC * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int B = p.B, S = p.cp.S, wps = S * S / 32;
  // per-row operands of the epilogue, ONE 16-byte LDS read per output value: {out_scale, offset + shift of field 0, of field 1,
  // element offset of the block's first cell in `fields` (bits)}; ownership words beside them (rows >= M own nothing)
  float4* lrow = reinterpret_cast<float4*>(lds + R * LDA);          // [R]
  uint32_t* lown = reinterpret_cast<uint32_t*>(lrow + R);           // [R][WPB]
  float* lg = reinterpret_cast<float*>(lown + R * WPB);             // [4] guard partial per wave
  const int ct = min((int)blockIdx.x * 4 + wave, a.n_coltiles - 1);
  const bool live = ((int)blockIdx.x * 4 + wave) < a.n_coltiles;
  const int m_first = (int)blockIdx.y * R, m_step = R * (int)gridDim.y;
  auto load_tile = [&](v4f (&x)[NA], int m_base) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, row = idx / Q, q = idx - row * Q;
      x[u] = *reinterpret_cast<const v4f*>(a.res + (int64_t)min(m_base + row, a.Mpad - 1) * LDR + 4 * q);
    }
  };
  // per-row operands of the epilogue (thread = row of the chunk): scale, offset + shift, ownership words
  struct RowOps { float rs, sa[C], sb[C]; uint32_t own[WPB]; int cs, b, y0, x0; bool in; };    // offset + shift = sa + sb, added when the row is written (the prefetch must not wait)
  auto load_rows = [&](RowOps& o, int m_base) {
    const int mr = m_base + min(tid, R - 1), m = min(mr, a.M - 1);
    o.in = mr < a.M;
    o.cs = m / B; o.b = m - o.cs * B;
    o.rs = a.row_scale[m];
    o.y0 = p.blk_y0x0[2 * o.b]; o.x0 = p.blk_y0x0[2 * o.b + 1];
#pragma unroll
    for (int w = 0; w < WPB; ++w) o.own[w] = p.ownbits[((int64_t)o.cs * B + o.b) * wps + (int)blockIdx.x * WPB + w];
  };
  // offset + shift of the row's block: requested AFTER the basis stream has been issued (first chunk), so that the sums'
  // wait does not sit in front of it
  auto load_sub = [&](RowOps& o) {
#pragma unroll
    for (int f = 0; f < C; ++f) {
      if (p.cf) {                                      // closed form: a0 + the B pair dots of this (case, field, block)
        o.sa[f] = p.cf_a0[((int64_t)o.cs * C + f) * B + o.b]; o.sb[f] = p.cf_dots[((int64_t)o.cs * C + f) * B + o.b];
      } else {
        o.sa[f] = p.offs[((int64_t)o.cs * C + f) * B + o.b]; o.sb[f] = p.shift[o.cs * C + f];
      }
    }
  };
  v4f x[NA];
  RowOps ro;
  load_tile(x, m_first);
  load_rows(ro, m_first);
  // guard flags of this solve (0, or NaN after a geometry mismatch): with the closed form there is no chain launch to fold
  // them into the shift, so every workgroup sums them itself.  The first 1024 (64 cases) are requested HERE, in front of the
  // basis stream: as a loop behind it they were a round trip of their own after everything else had landed.
  float gv0[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) gv0[u] = p.gflags[min(tid + 256 * u, p.n_gwaves - 1)];     // (unconditional: never null, >= 1 entry)
  __builtin_amdgcn_sched_barrier(0);
  float4 b[GD];
  const float4* bp = a.bpack + ((int64_t)ct * GD) * 64 + lane;
#pragma unroll
  for (int g = 0; g < GD; ++g) b[g] = stream_load(bp + g * 64);
  const int col = ct * 32 + i;
  const float mu = a.mean[col];
  __builtin_amdgcn_sched_barrier(0);
  load_sub(ro);
  float gpart = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) gpart += (p.cf && tid + 256 * u < p.n_gwaves) ? gv0[u] : 0.f;
  if (p.cf) {
    for (int k0 = tid + 1024; k0 < p.n_gwaves; k0 += 256 * 4) {      // more than 64 cases' worth of flags
      float gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) gv[u] = p.gflags[min(k0 + 256 * u, p.n_gwaves - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) gpart += (k0 + 256 * u < p.n_gwaves) ? gv[u] : 0.f;
    }
  }
  {
    const float gw = wave_sum(gpart);
    if (lane == 0) lg[wave] = gw;
  }
  // x6: this wave's basis slice split once into three bf16 planes (registers); reused by every row chunk
  x6_bf16x8 Bh[X6 ? LDR / 16 : 1], Bm[X6 ? LDR / 16 : 1], Bl[X6 ? LDR / 16 : 1];
  if constexpr (X6) {
#pragma unroll
    for (int st = 0; st < LDR / 16; ++st) {
      x6_bf16x4 h0, m0, l0, h1, m1, l1;
      psm_split3((f32x4){b[2 * st].x, b[2 * st].y, b[2 * st].z, b[2 * st].w}, h0, m0, l0);
      psm_split3((f32x4){b[2 * st + 1].x, b[2 * st + 1].y, b[2 * st + 1].z, b[2 * st + 1].w}, h1, m1, l1);
      Bh[st] = psm_cat4(h0, h1); Bm[st] = psm_cat4(m0, m1); Bl[st] = psm_cat4(l0, l1);
    }
  }
  const int px = col / C, f = col - px * C;
  const int pxl = px - (int)blockIdx.x * (128 / C);
  const int r = px / S, c = px - r * S;
  const uint32_t pix_off = (uint32_t)((r * p.Nx + c) * C + f);       // this lane's cell within a block's window, in elements
  const int own_w = pxl >> 5;
  const uint32_t own_bit = 1u << (pxl & 31);
  const float mu_r = psm_settled(mu);                  // (64 cases: 4.6 us per chunk for 0.6 us of MFMAs before this)
  // Row chunks.  The NEXT chunk's activation tile and row operands are requested as soon as the current tile sits in LDS and
  // land during its MFMAs and stores (they were a full exposed round trip per chunk: 64 cases are 4-5 chunks per workgroup).
  DSTAMP(1);
  int dchunk = 0;
  for (int m_base = m_first; m_base < m_end; m_base += m_step, ++dchunk) {
    if (m_base != m_first) __syncthreads();            // every wave is done with the previous chunk's tile and row operands
    DSTAMP(2 + 5 * dchunk);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, row = idx / Q, q = idx - row * Q;
      if constexpr (X6) {                            // exact three-way split, one bf16 plane each
        x6_bf16x4 vh, vm, vl;
        psm_split3(x[u], vh, vm, vl);
        __bf16* dst = reinterpret_cast<__bf16*>(lds) + row * LDX + 4 * q;
        *reinterpret_cast<x6_bf16x4*>(dst) = vh;
        *reinterpret_cast<x6_bf16x4*>(dst + R * LDX) = vm;
        *reinterpret_cast<x6_bf16x4*>(dst + 2 * R * LDX) = vl;
      } else if constexpr (BF) {
        pk_bf16x4 v;
        v[0] = (__bf16)x[u][0]; v[1] = (__bf16)x[u][1]; v[2] = (__bf16)x[u][2]; v[3] = (__bf16)x[u][3];
        *reinterpret_cast<pk_bf16x4*>(reinterpret_cast<__bf16*>(&lds[row * LDA]) + 4 * q) = v;
      } else {
        *reinterpret_cast<v4f*>(&lds[row * LDA + 4 * q]) = x[u];
      }
    }
    if (tid < R) {
      const uint32_t off = (uint32_t)(((int64_t)ro.cs * p.npix + (int64_t)ro.y0 * p.Nx + ro.x0) * C);
      lrow[tid] = make_float4(ro.rs, ro.sa[0] + ro.sb[0], ro.sa[C - 1] + ro.sb[C - 1], __uint_as_float(off));
#pragma unroll
      for (int w = 0; w < WPB; ++w) lown[tid * WPB + w] = ro.in ? ro.own[w] : 0u;
    }
    __syncthreads();
    DSTAMP(3 + 5 * dchunk);
    if (m_base + m_step < m_end) {                     // uniform
      load_tile(x, m_base + m_step);
      load_rows(ro, m_base + m_step);
      load_sub(ro);
    }
    __builtin_amdgcn_sched_barrier(0);
    DSTAMP(4 + 5 * dchunk);
    f32x16 acc[MTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
      acc[mt] = (f32x16){0};
      if constexpr (X6) {
        const __bf16* arow = reinterpret_cast<const __bf16*>(lds) + (mt * 32 + i) * LDX + 4 * h;
#pragma unroll
        for (int st = 0; st < LDR / 16; ++st) {
          const x6_bf16x8 ah = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 16 * st + 8));
          const x6_bf16x8 am = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + R * LDX + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + R * LDX + 16 * st + 8));
          const x6_bf16x8 al = psm_cat4(*reinterpret_cast<const x6_bf16x4*>(arow + 2 * R * LDX + 16 * st), *reinterpret_cast<const x6_bf16x4*>(arow + 2 * R * LDX + 16 * st + 8));
          acc[mt] = MFMA_X6(am, Bm[st], acc[mt]);
          acc[mt] = MFMA_X6(al, Bh[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bl[st], acc[mt]);
          acc[mt] = MFMA_X6(am, Bh[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bm[st], acc[mt]);
          acc[mt] = MFMA_X6(ah, Bh[st], acc[mt]);
        }
      } else if constexpr (BF) {
        const __bf16* arow = reinterpret_cast<const __bf16*>(&lds[(mt * 32 + i) * LDA]) + 8 * h;
#pragma unroll
        for (int g = 0; g < GD; ++g) {
          const pk_bf16x8 av = *reinterpret_cast<const pk_bf16x8*>(arow + 16 * g);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(pk_bf16x8, b[g]), acc[mt], 0, 0, 0);
        }
      } else {
        const float* arow = &lds[(mt * 32 + i) * LDA + 4 * h];
        float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll
        for (int g = 0; g < GD; ++g) {
          const float4 an = *reinterpret_cast<const float4*>(arow + 8 * (g + 1 < GD ? g + 1 : g));
          acc[mt] = MFMA32(av.x, b[g].x, acc[mt]);
          acc[mt] = MFMA32(av.y, b[g].y, acc[mt]);
          acc[mt] = MFMA32(av.z, b[g].z, acc[mt]);
          acc[mt] = MFMA32(av.w, b[g].w, acc[mt]);
          av = an;
        }
      }
    }
    DSTAMP(5 + 5 * dchunk);
    if (live) {
      const __amdgpu_buffer_rsrc_t frs = psm_store_rsrc(p.fields, p.field_bytes);
      const float gsum = (lg[0] + lg[1]) + (lg[2] + lg[3]);            // written before the barrier above
      auto paste = [&](auto buffered) {
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
#pragma unroll
          for (int rg = 0; rg < 16; ++rg) {
            const int rr = mt * 32 + acc_row(rg, h);
            const float4 ro4 = lrow[rr];
            const bool mine = (lown[rr * WPB + own_w] & own_bit) != 0u;
            const float val = (acc[mt][rg] + mu_r) * ro4.x - (C == 2 && f ? ro4.z : ro4.y) - gsum;
            if constexpr (decltype(buffered)::value) psm_store_if(frs, __float_as_uint(ro4.w) + pix_off, mine, val);
            else if (mine) p.fields[(size_t)(__float_as_uint(ro4.w) + pix_off)] = val;
          }
        }
      };
      if (p.field_bytes) paste(std::true_type{}); else paste(std::false_type{});      // uniform
    }
    DSTAMP(6 + 5 * dchunk);
  }
}

hipError_t psm_launch_decode_paste_batch(const PsmDecodeArgs& a, const PsmBoundBatchArgs& p, int c_out, hipStream_t st, int bf16) {
  if (a.ld_res > 128 || a.ld_res % 32 != 0 || a.Mpad % 32 != 0 || p.B > 4096 || p.B < 1 || a.M != p.B * p.n_cases) return hipErrorInvalidValue;
  if (c_out != 1 && c_out != 2) return hipErrorInvalidValue;
  const int nwg = (a.n_coltiles + 3) / 4;
  // at most 3 tiles of 32 rows per chunk: the 4-tile form of this kernel spills (acc + tile + row operands)
  // Row tiles per chunk (mtc <= 3: the 4-tile form spills) and row groups (grid.y).  With 256 or more column workgroups:
  // the most rows per pass over the weights.  With fewer (a single-field basis: 128) the rows are spread over up to
  // 512 / nwg groups, one chunk each where possible -- 8 cases x 9 blocks of a one-field model ran as 128 workgroups of 3
  // tiles (14.7 us), as 384 workgroups of one tile they take 10.6 us.  PSM_DECODE_MTC forces mtc (diagnostic).
  const int tiles = a.Mpad / 32, wpb = (128 / c_out) / 32;
  // One 32-row tile per chunk, the row tiles spread over up to `target / nwg` row groups (grid.y): measured against two and
  // three tiles per chunk (fewer passes over the basis slice, but 2-3x the LDS and registers per workgroup) at 8 ... 64
  // cases and both field counts -- 16 cases: 12.5 against 20.0 us, 64 cases: 30.2 against 48.8 us, U_to_gradP 8 cases: 24.1
  // against 36.8 us, 8 deltas cases: equal (tools/attic/decode_mtc_sweep.py).  PSM_DECODE_MTC / PSM_DECODE_WGS: diagnostic.
  static const int mtc_force = getenv("PSM_DECODE_MTC") ? atoi(getenv("PSM_DECODE_MTC")) : 0;
  static const int wg_target = getenv("PSM_DECODE_WGS") ? atoi(getenv("PSM_DECODE_WGS")) : 512;
  const int mtc = mtc_force ? std::min(std::max(mtc_force, 1), 3) : 1;
  const int groups = std::min((tiles + mtc - 1) / mtc, std::max(1, wg_target / nwg));
  const int R = mtc * 32;
  const bool x6 = a.x6 && !bf16;
  const size_t tile_floats = x6 ? (size_t)R * (a.ld_res + 4) * 3 / 2 : (size_t)R * (a.ld_res + 4);
  const size_t lds = (tile_floats + 4 * (size_t)R + (size_t)R * wpb + 4) * sizeof(float);
  if ((int64_t)p.n_cases * p.npix * c_out >= (int64_t)1 << 32) return hipErrorInvalidValue;   // cell offsets are 32-bit element counts
  const dim3 grid(nwg, groups);
#define DP(M_, C_, L_)                                                                                                          \
  do {                                                                                                                          \
    if (bf16) PSM_LAUNCH((psm_decode_paste_batch_kernel<M_, C_, L_, 1>), grid, dim3(256), lds, st, a, p, a.Mpad);      \
    else if (x6) PSM_LAUNCH((psm_decode_paste_batch_kernel<M_, C_, L_, 2>), grid, dim3(256), lds, st, a, p, a.Mpad);   \
    else PSM_LAUNCH((psm_decode_paste_batch_kernel<M_, C_, L_, 0>), grid, dim3(256), lds, st, a, p, a.Mpad);           \
  } while (0)
#define DPM(C_, L_)                                                                 \
  do {                                                                              \
    if (mtc == 3) DP(3, C_, L_); else if (mtc == 2) DP(2, C_, L_); else DP(1, C_, L_); \
  } while (0)
#define DPL(L_) do { if (c_out == 1) DPM(1, L_); else DPM(2, L_); } while (0)
  if (a.ld_res == 32) DPL(32); else if (a.ld_res == 64) DPL(64); else if (a.ld_res == 96) DPL(96); else DPL(128);
#undef DPL
#undef DPM
#undef DP
  return hipGetLastError();
}

// ---- bf16 handles on a bound geometry: the decode rounds `res` to bf16, which is not linear, so the strip dots
// cannot be folded through the head layer; they are taken from the rounded `res` itself in a small launch of their
// own (one wave per table row):  out[row] = scale * (bf16(res[b]) . G[row] + M[row]) / cnt[row]
// pair rows of the closed form (bind time): g2p[pair] = sum_e coef_e g2[src_e], c2p likewise
__global__ __launch_bounds__(256) void psm_pair_fold_kernel(PsmPairFoldArgs a) {
  const int pr = blockIdx.x;
  const int e0 = a.ptr[pr], e1 = a.ptr[pr + 1];
  for (int k = threadIdx.x; k < a.Kh; k += 256) {
    double acc = 0.0;
    for (int e = e0; e < e1; ++e) acc += (double)a.coef[e] * (double)a.g2[(int64_t)a.src[e] * a.Kh + k];
    a.g2p[(int64_t)pr * a.Kh + k] = (float)acc;
  }
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int e = e0; e < e1; ++e) acc += (double)a.coef[e] * (double)a.c2[a.src[e]];
    a.c2p[pr] = (float)acc;
  }
}
hipError_t psm_launch_pair_fold(const PsmPairFoldArgs& a, hipStream_t st) {
  if (a.n_pairs < 1 || a.Kh < 1) return hipErrorInvalidValue;
  PSM_LAUNCH(psm_pair_fold_kernel, dim3(a.n_pairs), dim3(256), 0, st, a);
  return hipGetLastError();
}

// table dots from any activation [rows][ld_act] (one wave per table row; d.Kh <= ld_act): introspection under the closed
// form (the strip means the chain would have consumed)
__global__ __launch_bounds__(256) void psm_act_dots_kernel(PsmDotsArgs d, const float* act, int ld_act, int round_bf16, int packed) {
  const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rc = min(row, d.n_rows - 1);
  const int blk = d.row_of[rc];
  float acc = 0.f;
  for (int k = lane; k < d.Kh; k += 64) {
    float x = packed ? act[psm_packed_offset(blk, k, ld_act >> 4)] : act[(int64_t)blk * ld_act + k];
    if (round_bf16) x = (float)(__bf16)x;
    acc += x * d.g2[(int64_t)rc * d.Kh + k];
  }
  const float tot = wave_sum(acc);
  if (lane == 0 && row < d.n_rows) d.out[row] = d.row_scale[blk] * (tot + d.c2[rc]) / d.cnt[rc];
}
hipError_t psm_launch_act_dots(const PsmDotsArgs& d, const float* act, int ld_act, int round_bf16, hipStream_t st, int packed) {
  if (d.n_rows < 1 || d.Kh < 1 || d.Kh > ld_act || (packed && ld_act % 16 != 0)) return hipErrorInvalidValue;
  PSM_LAUNCH(psm_act_dots_kernel, dim3((d.n_rows + 3) / 4), dim3(256), 0, st, d, act, ld_act, round_bf16, packed);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void psm_res_dots_kernel(PsmDotsArgs d, const float* res, int ld_res) {
  const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_dot_wgs = d.n_src > 1 ? d.n_rows : (d.n_rows + 3) / 4;
  if ((int)blockIdx.x >= n_dot_wgs) {                    // guard riders (PsmGuardArgs)
    psm_guard_wg<4>(d.guard, (int)blockIdx.x - n_dot_wgs, (int)(threadIdx.x >> 6), lane);
    return;
  }
  const int rc = min(row, d.n_rows - 1);
  if (d.n_src > 1) {                                     // closed form: one workgroup per row, wave w takes the source blocks w, w + 4, ...
    __shared__ float lsum[4];
    const int wave = threadIdx.x >> 6;
    const int rowl = (int)blockIdx.x, rl = min(rowl, d.n_rows - 1);
    const int cs = rl / d.rows_per_case;
    float acc = 0.f;
    for (int b0 = wave; b0 < d.n_src; b0 += 16) {
      float xv[4][2], gv[4][2], cc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int blk = min(b0 + 4 * t, d.n_src - 1);
        cc[t] = d.c2[(int64_t)rl * d.n_src + blk];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = min(lane + 64 * u, ld_res - 1);
          xv[t][u] = res[((int64_t)cs * d.n_src + blk) * ld_res + k];
          gv[t][u] = d.g2[((int64_t)rl * d.n_src + blk) * ld_res + k];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool on = b0 + 4 * t < d.n_src;
#pragma unroll
        for (int u = 0; u < 2; ++u) acc += (on && lane + 64 * u < ld_res) ? (float)(__bf16)xv[t][u] * gv[t][u] : 0.f;
        acc += (on && lane == 0) ? cc[t] : 0.f;
      }
    }
    const float tot = wave_sum(acc);
    if (lane == 0) lsum[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0 && rowl < d.n_rows) d.out[rowl] = d.row_scale[cs * d.n_src] * ((lsum[0] + lsum[1]) + (lsum[2] + lsum[3]));
    return;
  }
  const int blk = d.row_of[rc];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {                        // ld_res <= 128
    const int k = min(lane + 64 * u, ld_res - 1);
    const float x = (float)(__bf16)res[(int64_t)blk * ld_res + k];
    const float g = d.g2[(int64_t)rc * ld_res + k];
    acc += (lane + 64 * u < ld_res) ? x * g : 0.f;
  }
  const float tot = wave_sum(acc);
  if (lane == 0 && row < d.n_rows) d.out[row] = d.row_scale[blk] * (tot + d.c2[rc]) / d.cnt[rc];
}

hipError_t psm_launch_res_dots(const PsmDotsArgs& d, const float* res, int ld_res, hipStream_t st) {
  if (ld_res > 128 || ld_res < 1 || d.n_rows < 1) return hipErrorInvalidValue;
  const int nwg = (d.n_src > 1 ? d.n_rows : (d.n_rows + 3) / 4) + (d.guard.sdf ? d.guard.wg_count : 0);
  PSM_LAUNCH(psm_res_dots_kernel, dim3(nwg), dim3(256), 0, st, d, res, ld_res);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void psm_bind_copy_kernel(PsmBindArgs a) {   // un-folded tables: g2 = G, c2 = M
  const int row = blockIdx.x;
  for (int k = threadIdx.x; k < a.ld_out; k += 256) a.g2[(int64_t)row * a.ld_out + k] = (float)a.G[(int64_t)row * a.ld_out + k];
  if (threadIdx.x == 0) a.c2[row] = (float)a.Mrow[row];
}

hipError_t psm_launch_bind_unfolded(const PsmBindArgs& a, hipStream_t st) {
  if (a.ld_out > 512 || a.ld_out < 1 || (a.S * a.S) % 32 != 0) return hipErrorInvalidValue;
  const int rows = a.c_out * a.nst + a.c_out * a.B;
  PSM_LAUNCH(psm_bind_rows_kernel, dim3(rows), dim3(128), 0, st, a);
  PSM_LAUNCH(psm_bind_copy_kernel, dim3(rows), dim3(256), 0, st, a);
  PSM_LAUNCH(psm_bind_own_kernel, dim3((a.B * (a.S * a.S / 32) + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Ring stage-in: the grid of one ticket is pulled from (mapped) pinned host memory by the GPU itself -- 16-byte loads over
// PCIe, enough of them in flight to fill the link -- instead of a DMA-engine copy in front of the kernels: the whole
// ticket is then kernel nodes only (one cheap graph replay, no engine hand-over signals).  The same launch expands the
// per-case out_scale of the ticket (host, pinned) to the per-block-row scale the decode reads.
__global__ __launch_bounds__(256) void psm_stage_in_kernel(const float4* src, float4* dst, size_t n16, const float* tail_src,
                                                           float* tail_dst, int n_tail, const float* scale_host, float* row_scale,
                                                           int n_rows, int B) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  // four independent 16-byte loads per lane and round: 192 workgroups x 256 lanes x 64 B = 3 MiB in flight at most
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t < n_tail; t += 256) tail_dst[t] = tail_src[t];
    if (scale_host) for (int r = threadIdx.x; r < n_rows; r += 256) row_scale[r] = scale_host[r / B];
  }
}

hipError_t psm_launch_stage_in(const float* src_host, float* dst, size_t n_floats, const float* scale_host, float* row_scale,
                               int n_rows, int B, hipStream_t st) {
  const size_t n16 = n_floats / 4;
  const int n_tail = (int)(n_floats - 4 * n16);
  PSM_LAUNCH(psm_stage_in_kernel, dim3(192), dim3(256), 0, st, reinterpret_cast<const float4*>(src_host), reinterpret_cast<float4*>(dst), n16,
             src_host + 4 * n16, dst + 4 * n16, n_tail, scale_host, row_scale, n_rows, B);
  return hipGetLastError();
}
