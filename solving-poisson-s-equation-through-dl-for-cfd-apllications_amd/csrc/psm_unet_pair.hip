// psm_unet_pair.hip -- the two 3x3 convolutions of a U-Net level in ONE launch (bf16 mode, bf16 activations).
//
// At the 256^2 / 128^2 levels of the build-defined UNet-S (SURVEY.md §8 row a-conv; parity unpinned, see psm_unet.hip)
// a convolution is a few microseconds of MFMA work wrapped in a round trip of its whole activation through HBM and a
// launch boundary.  Here a workgroup owns an output tile of 30 x 14 pixels of the level's SECOND convolution and
// recomputes what it needs of the first one:
//   input tile  34 x 18 pixels (halo 2), read through the first convolution's source transform (raw image,
//               2x2 max-pool, 2x nearest-neighbour upsample ++ skip) into LDS,
//   mid tile    32 x 16 pixels (halo 1) = conv A + bias + ReLU, rounded to bf16, kept in LDS only
//               (pixels outside the image are written as 0: conv B's 'same' padding),
//   out tile    30 x 14 pixels = conv B + bias + ReLU (+ the fused linear 1x1 head).
// 32 mid columns = two MFMA pixel tiles exactly; the recomputed ring costs 22 % of conv A's MFMAs and saves the
// activation's HBM round trip and one launch.
//
// MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the first operand:  D[channel][pixel] -- lane l, register r holds
// channel 4*(l >> 4) + r of pixel l & 15, i.e. four consecutive channels of one pixel: one 8-byte bf16 store per lane
// into the mid tile or into the NHWC activation.  Second operand: lane l holds channels 8*(l >> 4) .. +7 of pixel l & 15.
// A wave owns R consecutive rows of one 16-pixel half: the R + 2 input rows of a tap column kx are read from LDS once
// and serve the three ky taps from registers ((R + 2) ds_read_b128 for 3R MFMAs per channel tile).
//   32-channel chunks: 64-byte LDS pixel, swizzled 16-byte slots (lds_slot, as in psm_unet.hip), 9 MFMAs per row.
//   16-channel chunks: 32-byte LDS pixel, unswizzled; lane groups 2,3 read the NEXT pixel, so one MFMA does the taps
//     kx and kx + 1 (weights of the pair stacked along k; the partner of kx = 2 has zero weights): 6 MFMAs per row.
#include "psm_unet.h"

#include <cstdlib>
#include <type_traits>

// Diagnostic stamps (100 MHz wall clock) of the first, the middle and the last workgroup -- compiled only with -DPSM_STAMPS.
#ifdef PSM_STAMPS
__device__ unsigned long long g_pair_stamps[64];
// slot = 9 * iteration + k of workgroup PSM_STAMP_WG (default 0): the tile loop of one persistent workgroup
#ifndef PSM_STAMP_WG
#define PSM_STAMP_WG 0
#endif
#define PSTAMP(k) do { if (blockIdx.x == PSM_STAMP_WG && threadIdx.x == 0 && 9 * g_it + (k) < 63) g_pair_stamps[9 * g_it + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PSTAMP_ENTRY() do { if (blockIdx.x == PSM_STAMP_WG && threadIdx.x == 0) g_pair_stamps[63] = __builtin_amdgcn_s_memrealtime(); } while (0)   // kernel entry
#define PSTAMP_PRO(k) do { if (blockIdx.x == PSM_STAMP_WG && threadIdx.x == 0) g_pair_stamps[56 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // prologue, k < 7
hipError_t psm_unet_pair_read_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), sizeof(g_pair_stamps)); }
#else
#define PSTAMP(k) do { } while (0)
#define PSTAMP_ENTRY() do { } while (0)
#define PSTAMP_PRO(k) do { } while (0)
hipError_t psm_unet_pair_read_stamps(unsigned long long* out) { for (int i = 0; i < 64; ++i) out[i] = 0; return hipSuccess; }
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TX = PSM_PAIR_TX, TY = PSM_PAIR_TY;       // output tile
constexpr int MH = TY + 2;                               // mid tile rows (32 columns)
constexpr int IH = TY + 4;                               // input tile rows
constexpr int P32 = 40;                                  // pitch (pixels) of a 64-byte-pixel tile (34 used); a multiple of 8: the slot
                                                         // swizzle then depends on the column only and rows are immediate offsets
constexpr int P16 = 35;                                  // pitch of a 32-byte-pixel tile: the paired tap of kx = 2 reads one pixel further
constexpr int T32_BYTES = P32 * IH * 64;                 // 46080
constexpr int T16_BYTES = P16 * IH * 32;                 // 20160
constexpr int M32_BYTES = P32 * MH * 64;                 // 40960
constexpr int M16_BYTES = P16 * MH * 32;                 // 17920

// diagnostic builds (-DPSM_PAIR_EXP=n, results wrong on purpose): 1 no MFMAs, 2 no tile staging, 3 neither; 4 the staging requests from
// trivially cheap addresses (upper bound of an address-arithmetic diet), 8 requests as shipped but no LDS writes, 12 both, 16 the fused
// head's stores dropped
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 1)
#define MFMA_BF(w, x, c) (c)
#else
#define MFMA_BF(w, x, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((w), (x), (c), 0, 0, 0)
#endif
// level-0 decoder pair: the next tile's ten requests spread over conv A's MFMA steps, two per step (1), or issued in one go
// before them (0).  Measured A/B on one box (round 4, profiles/archive/r04_conv_experiments.txt (5g)): 20.50 / 19.80 us spread against
// 20.06 / 20.00 us in one go at 8 cases, 13.30 / 13.16 against 13.46 / 13.42 us at 512 x 512 x 1 -- the address path is not what
// the tile waits for either.  Off; -DPSM_PAIR_SPREAD=1 builds it (parity tests green in both forms).
#ifndef PSM_PAIR_SPREAD
#define PSM_PAIR_SPREAD 0
#endif

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the NEXT tile's
// global requests, which are meant to stay in flight across the barriers
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// byte offset of 16-byte channel group `grp` (0..3) of pixel P in a 64-byte-pixel tile (see psm_unet.hip, lds_slot)
__device__ __forceinline__ int slot64(int P, int grp) { return P * 64 + 16 * ((grp + ((P >> 1) & 2)) & 3); }

// element-wise maximum of two groups of 8 packed bf16 (16 bytes each).  The 2 x 2 max-pool only ever reads outputs of ReLU'd convolutions
// (every 3x3 layer has one; the linear head feeds no pool): NON-NEGATIVE values, whose bf16 bit patterns order like signed 16-bit integers
// (-0 below +0, a positive NaN above everything, so it propagates like np.maximum) -- four v_pk_max_i16 instead of the 36 shift / mask /
// float-max / pack instructions of the float form, three times per staged piece (round 6: the pool staging of the encoder launches is
// vector-instruction bound).
__device__ __forceinline__ f32x4 bf16x8_max(f32x4 a, f32x4 b) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
}

// four floats -> four bf16 (RNE): two v_cvt_pk_bf16_f32 (element-wise scalar conversions compile to four plus two v_perm)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 pack4(f32x4 v) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  u32x2 r;
  r[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
  r[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2));
  return r;
}

// ---- staging, row-wise --------------------------------------------------------------------------------------------
// A wave64 vector instruction costs 4 issue cycles, so the address arithmetic of the staging is what these kernels spend
// their non-MFMA time on.  The mapping keeps it to a handful of instructions per tile: wave w stages tile rows w, w + 4,
// ... (the row is wave-uniform: row pointer, row validity and the LDS row offset are scalar), and a lane keeps ONE column
// position per instruction slot (computed once per tile): PPI = 64 / G pixels per instruction, KM instructions per row
// for the first MAINPX columns; the few remaining columns of all rows ("side block") are one more instruction over the
// workgroup's 256 threads.  The LDS layout is the one the convolutions read, whatever the mapping.
//   PXB: bytes per LDS pixel (32: 16 channels, unswizzled; 64: 32 channels, swizzled slots);  NCOL x NROW staged pixels;
//   NQ = 4: the source has twice the resolution and the piece is the element-wise maximum of its 2x2 pixels (max-pool).
// issue(): requests only (clamped addresses), ok = which pieces lie inside the image;  write(): combine, zero, store.
template <int PXB, int NCOL, int NROW, int PITCH, int NQ>
struct Stage {
  static constexpr int G = PXB / 16, PPI = 64 / G, MAINPX = (NCOL / PPI) * PPI, KM = MAINPX / PPI, RW = (NROW + 3) / 4;
  static constexpr int SIDEW = NCOL - MAINPX, NSIDE = SIDEW * G * NROW, NP = RW * KM + (NSIDE > 0 ? 1 : 0);
  static_assert(NSIDE <= 256 && NP <= 16, "one side instruction, ok bits in 16");
  static __device__ __forceinline__ int lds_off(int r, int c, int g) {
    return PXB == 64 ? r * (PITCH * 64) + slot64(c, g) : (r * PITCH + c) * 32 + g * 16;
  }
  // src: the case's source image, Hs x Ws pixels of cpx channels (NQ = 4: 2Hs x 2Ws); (ys0, xs0): source pixel of tile (0, 0)
  // J0, JN: the row iterations of this call (registers: a max-pool chunk is staged in two halves); SIDE: with the side block
  template <int J0 = 0, int JN = RW, bool SIDE = (NSIDE > 0)>
  static __device__ __forceinline__ void issue(f32x4 (&v)[JN * KM + (SIDE ? 1 : 0)][NQ], unsigned& ok, const unsigned short* src, int cpx, int cb,
                                               int Hs, int Ws, int Ps, int ys0, int xs0, int wave, int lane, int tid) {
    constexpr int NV = JN * KM + (SIDE ? 1 : 0);
    ok = 0;
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
    return;
#endif
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP == 4 || PSM_PAIR_EXP == 12)
    {   // upper bound of an address-arithmetic diet: the same number of requests, from trivially cheap (wrong) addresses
      const char* q = reinterpret_cast<const char*>(src) + tid * 16 + (ys0 & 1) * 64;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < NQ; ++e) v[i][e] = *reinterpret_cast<const f32x4*>(q + 4096 * (i * NQ + e));
      ok = ~0u;
      return;
    }
#endif
    const int M = NQ == 4 ? 2 : 1;                           // source pixels per tile pixel and direction
    const int64_t row_bytes = (int64_t)Ps * cpx * 2;         // one source row (Ps: the source tensor's pitch in ITS pixels)
    // (round 6) the sources of the 32-channel pairs are ZERO-HALOED bf16 tensors (psm_unet_api.cpp, act_layout: 4 pixels left / top, 64 / 32
    // right / bottom): the 'same' padding, the last tile's overhang and a max-pool's doubled extent are read from memory like any other
    // pixel -- no clamps, no predicates, and write() below needs no selects (Hs, Ws stay in the signature for the callers)
    (void)Hs; (void)Ws;
    int xoff[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int c = PPI * k + lane / G, g = lane % G, x = xs0 + c;
      xoff[k] = (M * x * cpx + cb + 8 * g) * 2;
    }
    ok = ~0u;
#pragma unroll
    for (int jj = 0; jj < JN; ++jj) {
      const int j = jj, r = min(wave + 4 * (J0 + jj), NROW - 1), y = ys0 + r;          // uniform
      const char* rowp = reinterpret_cast<const char*>(src) + (int64_t)(M * y) * row_bytes;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        const char* q = rowp + xoff[k];
        v[j * KM + k][0] = *reinterpret_cast<const f32x4*>(q);
        if (NQ == 4) {
          v[j * KM + k][NQ == 4 ? 1 : 0] = *reinterpret_cast<const f32x4*>(q + cpx * 2);
          v[j * KM + k][NQ == 4 ? 2 : 0] = *reinterpret_cast<const f32x4*>(q + row_bytes);
          v[j * KM + k][NQ == 4 ? 3 : 0] = *reinterpret_cast<const f32x4*>(q + row_bytes + cpx * 2);
        }
      }
    }
    if (SIDE) {
      const int q = min(tid, NSIDE - 1), r = q / (SIDEW * G), rem = q - r * (SIDEW * G), c = MAINPX + rem / G, g = rem % G;
      const int y = ys0 + r, x = xs0 + c;
      const char* qp = reinterpret_cast<const char*>(src) + (int64_t)(M * y) * row_bytes + (M * x * cpx + cb + 8 * g) * 2;
      v[NV - 1][0] = *reinterpret_cast<const f32x4*>(qp);
      if (NQ == 4) {
        v[NV - 1][NQ == 4 ? 1 : 0] = *reinterpret_cast<const f32x4*>(qp + cpx * 2);
        v[NV - 1][NQ == 4 ? 2 : 0] = *reinterpret_cast<const f32x4*>(qp + row_bytes);
        v[NV - 1][NQ == 4 ? 3 : 0] = *reinterpret_cast<const f32x4*>(qp + row_bytes + cpx * 2);
      }
    }
  }
  // ONE piece of issue() (whole tile: J0 = 0, JN = RW, with the side block; NQ == 1), so that a kernel can spread the requests
  // of the next tile over the MFMA steps of the current one: a wave-wide 16-byte load holds the CU's address path for 16
  // cycles, and ten of them issued in one go by eight waves in phase keep every wave -- and the MFMAs behind the loads in
  // program order -- for up to 0.6 us per tile.  I < RW * KM: row iteration I / KM, column slot I % KM; I == RW * KM: side block.
  // The first piece clears ok.
  template <int I>
  static __device__ __forceinline__ void issue_piece(f32x4 (&v)[NP][NQ], unsigned& ok, const unsigned short* src, int cpx, int cb,
                                                     int Hs, int Ws, int Ps, int ys0, int xs0, int wave, int lane, int tid) {
    static_assert(NQ == 1 && I >= 0 && I < NP, "same-resolution pieces of a whole tile");
    if (I == 0) ok = ~0u;                                  // zero-haloed sources: nothing to clamp, nothing to zero (see issue())
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
    return;
#endif
    (void)Hs; (void)Ws;
    const int64_t row_bytes = (int64_t)Ps * cpx * 2;
    if constexpr (I < RW * KM) {
      constexpr int jj = I / KM, k = I % KM;
      const int c = PPI * k + lane / G, g = lane % G, x = xs0 + c;
      const int xoff = (x * cpx + cb + 8 * g) * 2;
      const int r = min(wave + 4 * jj, NROW - 1), y = ys0 + r;          // uniform
      const char* rowp = reinterpret_cast<const char*>(src) + (int64_t)y * row_bytes;
      v[I][0] = *reinterpret_cast<const f32x4*>(rowp + xoff);
    } else {
      const int q = min(tid, NSIDE - 1), r = q / (SIDEW * G), rem = q - r * (SIDEW * G), c = MAINPX + rem / G, g = rem % G;
      const int y = ys0 + r, x = xs0 + c;
      const char* qp = reinterpret_cast<const char*>(src) + (int64_t)y * row_bytes + (x * cpx + cb + 8 * g) * 2;
      v[I][0] = *reinterpret_cast<const f32x4*>(qp);
    }
  }
  // interior (uniform): every staged pixel lies inside the image -- no selects
  template <int J0 = 0, int JN = RW, bool SIDE = (NSIDE > 0)>
  static __device__ __forceinline__ void write(char* tile, const f32x4 (&v)[JN * KM + (SIDE ? 1 : 0)][NQ], unsigned ok, bool interior, int wave, int lane, int tid) {
    constexpr int NV = JN * KM + (SIDE ? 1 : 0);
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
    return;
#endif
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP == 8 || PSM_PAIR_EXP == 12)
    {   // requests as shipped, but nothing written: one store of a value that depends on every request (so that none is dropped)
      f32x4 t = v[0][0];
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < NQ; ++e) t += v[i][e];
      if (t[0] == 12345.678f) *reinterpret_cast<f32x4*>(tile) = t;
      return;
    }
#endif
    auto piece = [&](int i) {
      f32x4 t = v[i][0];
      if (NQ == 4) t = bf16x8_max(bf16x8_max(v[i][0], v[i][NQ == 4 ? 1 : 0]), bf16x8_max(v[i][NQ == 4 ? 2 : 0], v[i][NQ == 4 ? 3 : 0]));
      (void)ok; (void)interior;                            // zero-haloed sources: out-of-image pixels arrive as zeros
      return t;
    };
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int r = min(wave + 4 * (J0 + j), NROW - 1);
#pragma unroll
      for (int k = 0; k < KM; ++k)
        *reinterpret_cast<f32x4*>(tile + lds_off(r, PPI * k + lane / G, lane % G)) = piece(j * KM + k);
    }
    if (SIDE) {
      const int q = min(tid, NSIDE - 1), r = q / (SIDEW * G), rem = q - r * (SIDEW * G);
      *reinterpret_cast<f32x4*>(tile + lds_off(r, MAINPX + rem / G, rem % G)) = piece(NV - 1);
    }
  }
};
constexpr int PL = 24;                                    // LDS pitch of the low-resolution tile (17 used)
constexpr int LOW_BYTES = (IH / 2) * PL * 64;             // 13824
typedef Stage<64, 17, IH / 2, PL, 1> StLow;               // upsample source at its own resolution, 32 channels (tile origins are even)
typedef Stage<32, P16, IH, P16, 1> St16;                  // 16 channels, same resolution
typedef Stage<64, 34, IH, P32, 1> St32;                   // 32 channels, same resolution
typedef Stage<32, P16, IH, P16, 4> StPool16;              // 16 channels through the 2x2 max-pool
typedef Stage<64, 34, IH, P32, 4> StPool32;               // 32 channels through the 2x2 max-pool

// ---- one chunk of a convolution on an LDS tile --------------------------------------------------------------------
// acc[m][nt]: row r0 + m, channel tile nt of the wave's 16-pixel half.  A chunk is S steps (tap columns); per step the
// NX input rows and the 3 NT weight fragments are read from LDS one step ahead of the MFMAs that use them (two register
// stages; nothing moves across the sched_barriers, so the live set stays at two stages).
//   xload(s, i): operand of step s, input-row slot i;   wload(s, ky, nt): weight fragment;   HALF: slot of row j is j >> 1
struct NoFill { __device__ __forceinline__ void operator()(int) const {} };
//   fill(i): up to two slices of unrelated work per step (i = 2 s after the ky = 0 MFMAs, 2 s + 1 after the ky = 1 ones), pinned
//   between the MFMA groups: what is issued there runs in the shadow of the MFMAs already in the pipe
template <int NT, int R, int NX, int S, bool HALF, bool DB, typename XLoad, typename WLoad, typename Fill = NoFill>
__device__ __forceinline__ void conv_steps(XLoad xload, WLoad wload, f32x4 (&acc)[R][NT], Fill fill = Fill()) {
  bf16x8 xb[DB ? 2 : 1][NX], wb[DB ? 2 : 1][3 * NT];
  auto load = [&](int s, int st) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xb[st][i] = xload(s, i);
#pragma unroll
    for (int k = 0; k < 3 * NT; ++k) wb[st][k] = wload(s, k / NT, k % NT);
  };
  if (DB) load(0, 0);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (DB) { if (s + 1 < S) load(s + 1, (s + 1) & 1); }
    else load(s, 0);                     // one stage (register budget): the other wave of the SIMD covers the read latency
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < R; ++m)
          acc[m][nt] = MFMA_BF(wb[DB ? (s & 1) : 0][ky * NT + nt], xb[DB ? (s & 1) : 0][HALF ? (m + ky) >> 1 : m + ky], acc[m][nt]);
      if constexpr (!std::is_same<Fill, NoFill>::value) {
        if (ky < 2) {
          __builtin_amdgcn_sched_barrier(0);
          fill(2 * s + ky);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// 32-channel chunk, 64-byte pixels (swizzled slots), pitch P32: step = tap column kx; fragments [kx][ky][nt]
template <int NT, int R, typename WGet>
__device__ __forceinline__ void conv32(const char* tile, int r0, int xcol, int kq, WGet wget, f32x4 (&acc)[R][NT]) {
  conv_steps<NT, R, R + 2, 3, false, NT == 1>(
      [&](int kx, int i) { return *reinterpret_cast<const bf16x8*>(tile + r0 * (P32 * 64) + slot64(xcol + kx, kq) + i * (P32 * 64)); },
      [&](int kx, int ky, int nt) { return wget((kx * 3 + ky) * NT + nt); }, acc);
}
// the same on a HALF-resolution tile (2x nearest-neighbour upsample done by the addressing): pixel (r, c) of the input
// tile is low-resolution pixel (r >> 1, c >> 1), pitch PL; r0 is even, so rows r0 + 2j and r0 + 2j + 1 share one read
template <int NT, int R, typename WGet, typename Fill = NoFill>
__device__ __forceinline__ void conv32_up(const char* tile, int r0, int xcol, int kq, WGet wget, f32x4 (&acc)[R][NT], Fill fill = Fill()) {
  static_assert((R & 1) == 0, "row pairs");
  conv_steps<NT, R, (R + 2) / 2, 3, true, NT == 1>(
      [&](int kx, int i) { return *reinterpret_cast<const bf16x8*>(tile + (r0 >> 1) * (PL * 64) + slot64((xcol + kx) >> 1, kq) + i * (PL * 64)); },
      [&](int kx, int ky, int nt) { return wget((kx * 3 + ky) * NT + nt); }, acc, fill);
}
// 16-channel chunk, 32-byte pixels, pitch P16: step = tap pair (kx = 2s, 2s + 1); fragments [s][ky][nt]
template <int NT, int R, bool DB = (NT == 1), typename WGet, typename Fill = NoFill>
__device__ __forceinline__ void conv16(const char* tile, int r0, int xcol, int kq, WGet wget, f32x4 (&acc)[R][NT], Fill fill = Fill()) {
  conv_steps<NT, R, R + 2, 2, false, DB>(
      [&](int s, int i) { return *reinterpret_cast<const bf16x8*>(tile + ((r0 + i) * P16 + xcol + 2 * s) * 32 + kq * 16); },
      [&](int s, int ky, int nt) { return wget((s * 3 + ky) * NT + nt); }, acc, fill);
}

// Persistent workgroups: workgroup w runs tiles it = 0, 1, ... of its own sequence.  Consecutive workgroup ids go to
// different XCDs, so the workgroups of one XCD (w % 8) share a contiguous range of tiles -- neighbouring tiles meet in
// that XCD's L2, where they share their halo.  -1: no such tile.
__device__ __forceinline__ int tile_of(int it, int total) {
  const int G = gridDim.x, w = blockIdx.x;
  if ((total & 7) == 0 && (G & 7) == 0) {
    const int per = total >> 3, idx = (w >> 3) + it * (G >> 3);
    return idx < per ? (w & 7) * per + idx : -1;
  }
  const int t = w + it * G;
  return t < total ? t : -1;
}
// The same descriptor computed in the kernel (two integer divisions, ~100 scalar instructions): for a workgroup's FIRST tile, whose
// table entry would be a cold fetch in front of everything else (measured: the 128^2 pairs, one tile per workgroup, 13 -> 15 us
// and 19 -> 21 us with the table alone).  Later tiles come from the table, requested a whole tile ahead.
template <int KIND>
__device__ __forceinline__ PsmPairTile tile_calc(const PsmPairArgs& a, int t) {
  const int per_case = a.tiles_x * a.tiles_y;
  const int cs = t / per_case, r = t - cs * per_case, by = r / a.tiles_x, bx = r - by * a.tiles_x;
  PsmPairTile d;
  d.y0 = by * TY; d.x0 = bx * TX; d.cs = cs;
  d.pix = (cs * a.H + d.y0) * a.W + d.x0;
  d.flags = (d.x0 >= 2 && d.x0 + 33 <= a.W && d.y0 >= 2 && d.y0 + 16 <= a.H) ? 1 : 0;
  d.offo = (int)((cs * a.out_case + ((int64_t)d.y0 * a.PO + d.x0) * a.cm) * 2);
  d.off0 = d.off1 = 0;
  if (KIND == PSM_PAIR_UPCAT) {
    d.off0 = (int)((cs * a.in0_case + ((int64_t)(d.y0 / 2 - 1) * a.P0 + (d.x0 / 2 - 1)) * a.c0) * 2);
    d.off1 = (int)((cs * a.in1_case + ((int64_t)(d.y0 - 2) * a.P1 + (d.x0 - 2)) * a.c1) * 2);
  } else if (KIND == PSM_PAIR_POOL) {
    d.off0 = (int)((cs * a.in0_case + ((int64_t)(2 * d.y0 - 4) * a.P0 + (2 * d.x0 - 4)) * a.c0) * 2);
  }
  return d;
}
// tile t's descriptor (host-built, psm_unet.h): a wave-uniform index -> scalar loads
// (constant address space + a readfirstlane'd index: s_load_dwordx8; through the generic pointer the compiler issued eight
// vector loads and a readfirstlane each, and their vmcnt waits landed in the MFMA stream)
__device__ __forceinline__ PsmPairTile tile_desc(const PsmPairArgs& a, int t) {
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  const auto* p = (const __attribute__((address_space(4))) i32x8*)a.tiles;
  const i32x8 v = p[__builtin_amdgcn_readfirstlane(t)];
  return PsmPairTile{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
}

// ---- epilogues ------------------------------------------------------------------------------------------------------
// The bias is the accumulator's initial value.  ReLU on packed bf16: max with 0 as signed 16-bit integers (a negative float
// has the sign bit set; rounding is sign-symmetric, so rounding first changes nothing).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 pack4_relu(f32x4 v) {
  const u32x2 h = pack4(v);
  const s16x2 z = {0, 0};
  const unsigned lo = h[0], hi = h[1];                     // (element-wise updates of h in place were miscompiled: both halves from h[0])
  const s16x2 a = __builtin_elementwise_max(__builtin_bit_cast(s16x2, lo), z), b = __builtin_elementwise_max(__builtin_bit_cast(s16x2, hi), z);
  u32x2 r;
  r[0] = __builtin_bit_cast(unsigned, a); r[1] = __builtin_bit_cast(unsigned, b);
  return r;
}
// conv A: ReLU, zero outside the image (border tiles only), bf16 into the mid tile (LDS).  KEEP: also to HBM, every pixel
// by the tile that owns it (introspection).
template <int NT, int R, bool KEEP>
__device__ __forceinline__ void mid_epilogue(const PsmPairArgs& a, char* mid, int cs, int y0, int x0, int r0, int xh, int lane,
                                             bool interior, f32x4 (&acc)[R][NT]) {
  const int px = lane & 15, kq = lane >> 4;
  const int col = 16 * xh + px, x = x0 - 1 + col;
  const bool xok = x >= 0 && x < a.W;
  char* base[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    base[nt] = mid + (NT == 1 ? (r0 * P16 + col) * 32 + kq * 8 : r0 * (P32 * 64) + slot64(col, 2 * nt + (kq >> 1)) + (kq & 1) * 8);
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const int y = y0 - 1 + r0 + m;
    const bool ok = xok && y >= 0 && y < a.H;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      u32x2 h = pack4_relu(acc[m][nt]);
      if (!interior) { h[0] = ok ? h[0] : 0u; h[1] = ok ? h[1] : 0u; }
      *reinterpret_cast<u32x2*>(base[nt] + m * (NT == 1 ? P16 * 32 : P32 * 64)) = h;
      if (KEEP && ok && r0 + m >= 1 && r0 + m <= TY && col >= 1 && col <= TX)
        *reinterpret_cast<u32x2*>(a.mid_out + (int64_t)cs * a.out_case + ((int64_t)y * a.PO + x) * (16 * NT) + 16 * nt + 4 * kq) = h;
    }
  }
}
// Activation stores are WRITE-THROUGH (sc1) where PSM_WT_STORES is on: the next launch runs on other XCDs, whose L2s are not
// coherent with this one's, so every dirty line is written back at the end of the kernel anyway -- as one burst behind the last
// workgroup (MI355X_MICROARCH.md, 'boundary': + B / 6 TB/s for B dirty bytes: 2.8 us behind the stem pair's 16.8 MB).  Written
// through, the lines leave L2 while the kernel still computes.  (Inline asm: hipcc has no store builtin with cache bits for a flat
// address; an 8-byte store needs no trailing wait state, the compiler does not count it in vmcnt -- its own counted waits then
// over-wait, never under-wait.)
#ifndef PSM_WT_STORES
#define PSM_WT_STORES 1
#endif
__device__ __forceinline__ void store_act8(void* p, u32x2 v) {
#if PSM_WT_STORES
  asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<u32x2*>(p) = v;
#endif
}
// conv B: ReLU, bf16 NHWC store (STORE), fused linear 1x1 head (HEAD; headw in LDS: [16][head_cout], biases at [256]).
// The head needs the sum over a pixel's 16 channels = the four lanes l, l ^ 16, l ^ 32, l ^ 48 after the in-lane part.
// Four rows at a time: v_permlane32_swap of two rows' partial sums + one add leaves row A's lane-pair sums in lanes
// 0-31 and row B's in lanes 32-63; v_permlane16_swap of two such registers + one add finishes four rows in one register
// (16-lane group g holds row {0, 2, 1, 3}[g]) -- 6 instructions and ONE store for four rows.
template <int NT, int R, bool STORE, bool HEAD>
__device__ __forceinline__ void out_epilogue(const PsmPairArgs& a, char* otile, float* htile, int y0, int x0, int r0, int xh, int lane,
                                             const float* headw, f32x4 (&acc)[R][NT]) {
  // otile: pixel (y0, x0) of this case in a.out (bytes);  htile: the same pixel in a.head_out -- both wave-uniform; a lane adds 32-bit offsets
  static_assert(!HEAD || NT == 1, "the fused head reads one channel tile");
  const int px = lane & 15, kq = lane >> 4;
  const int col = 16 * xh + px, x = x0 + col;
  const bool xok = col < TX && x < a.W;
  if constexpr (STORE) {
    const unsigned loff = (unsigned)(((r0 * a.PO + col) * (16 * NT) + 4 * kq) * 2);
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const bool ok = xok && y0 + r0 + m < a.H;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (ok) store_act8(otile + loff + (unsigned)(m * a.PO * (16 * NT) * 2) + nt * 32, pack4_relu(acc[m][nt]));
    }
  }
  if constexpr (HEAD) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int grp_row = ((kq & 1) << 1) | (kq >> 1);        // row (within a group of four) whose total this lane group ends up with
    for (int o = 0; o < a.head_cout; ++o) {
      f32x4 hw;
#pragma unroll
      for (int j = 0; j < 4; ++j) hw[j] = headw[(4 * kq + j) * a.head_cout + o];
      const float hb = headw[256 + o];
      float part[(R + 3) / 4 * 4];
#pragma unroll
      for (int m = 0; m < (R + 3) / 4 * 4; ++m) {
        if (m < R) {
          const f32x4 v = acc[m][0];
          part[m] = fmaxf(v[0], 0.f) * hw[0] + fmaxf(v[1], 0.f) * hw[1] + fmaxf(v[2], 0.f) * hw[2] + fmaxf(v[3], 0.f) * hw[3];
        } else part[m] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < (R + 3) / 4; ++q) {
        const u32x2v s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[4 * q]), __float_as_uint(part[4 * q + 1]), false, false);
        const u32x2v s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[4 * q + 2]), __float_as_uint(part[4 * q + 3]), false, false);
        const float u01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);       // lanes 0-31: row 0 (l + l^32), lanes 32-63: row 1
        const float u23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
        const u32x2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(u01), __float_as_uint(u23), false, false);
        const float tot = __uint_as_float(t[0]) + __uint_as_float(t[1]) + hb;      // group g: row {0, 2, 1, 3}[g] of this group of four
        const int m = 4 * q + grp_row, y = y0 + r0 + m;
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP == 16)
        if (tot == 12345.678f)                     // diagnostic: the head's result computed, (almost) never stored
#endif
        if (xok && m < R && y < a.H) htile[(unsigned)(((r0 + m) * a.W + col) * a.head_cout + o)] = tot;
      }
    }
  }
}
// the two tile pointers from a descriptor (one 64-bit scalar add each)
__device__ __forceinline__ char* out_tile(const PsmPairArgs& a, const PsmPairTile& d) { return reinterpret_cast<char*>(a.out) + d.offo; }
__device__ __forceinline__ float* head_tile(const PsmPairArgs& a, const PsmPairTile& d) { return a.head_out + (int64_t)d.pix * a.head_cout; }
__device__ __forceinline__ void load_head(const PsmPairArgs& a, float* headw, int tid) {
  if (a.head_w) {
    if (tid < 16 * a.head_cout) headw[tid] = a.head_w[tid];
    if (tid < a.head_cout) headw[256 + tid] = a.head_b[tid];
  }
}
__device__ __forceinline__ void zero_mid_pad16(char* mid, int tid) {          // columns 32..34 of the 16 mid rows
  if (tid < MH * 3) {
    const int r = tid / 3, c = 32 + tid - 3 * r;
    f32x4* p = reinterpret_cast<f32x4*>(mid + (r * P16 + c) * 32);
    p[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; p[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void zero_mid_pad32(char* mid, int tid) {          // columns 32, 33
  if (tid < MH * 2 * 4) {
    const int pix = tid >> 2, r = pix >> 1, c = 32 + (pix & 1);
    *reinterpret_cast<f32x4*>(mid + (r * P32 + c) * 64 + (tid & 3) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
// a lane-linear copy of n_pieces KiB (weight fragments as the host packed them) into LDS, piece i by wave i % 4
__device__ __forceinline__ void dma_copy(char* dst, const void* src, int n_pieces, int wave, int lane) {
  const char* s = reinterpret_cast<const char*>(src) + 16 * lane;
  for (int i = wave; i < n_pieces; i += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}
// the issuing wave's LDS-DMA pieces have landed (vmcnt counts them like loads); the barrier behind it publishes them to the others
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ====================================================================================================================
// level 0, encoder: raw image (C0 = 3 or 4 float32 channels) -> 16 -> 16.  conv A flattens k = tap*C0 + channel and
// pads it to KS steps of 32; its operand is gathered from the bf16 image tile with per-lane fixed offsets.
// ====================================================================================================================
template <int C0, bool KEEP>
__global__ __launch_bounds__(256, 2) void psm_pair_stem16_kernel(PsmPairArgs a) {
  constexpr int K = 9 * C0, KS = (K + 31) / 32;
  constexpr int PI = 34;                                  // pitch of the image tile (pixels)
  constexpr int RW = (IH + 3) / 4;                        // image rows per wave
  constexpr int RV = PI * C0, KR = (RV + 63) / 64;        // values per row, load instructions per row
  __shared__ __attribute__((aligned(16))) unsigned short img[IH * RV + 64];
  __shared__ __attribute__((aligned(16))) char mid[M16_BYTES];
  __shared__ __attribute__((aligned(16))) bf16x8 wbl[6 * 64];            // conv B's fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  PSTAMP_ENTRY();
  const bf16x8* wbf = wbl + lane;
  int off[KS][8];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + 8 * kq + j, kc = min(k, K - 1), tap = kc / C0, ci = kc - tap * C0;
      const int ky = tap / 3, kx = tap - 3 * ky;
      off[s][j] = (16 * xh + px) * C0 + (k < K ? (ky * PI + kx) * C0 + ci : 0);      // k >= K: any finite value, its weight is zero
    }
  zero_mid_pad16(mid, tid);
  // image staging, row-wise: wave w takes rows w, w + 4, ...; a lane keeps KR value positions of the row
  float ev[RW][KR];
  unsigned evok = 0;
  auto issue = [&](const PsmPairTile& t) {
    const float* in0 = reinterpret_cast<const float*>(a.in0) + (int64_t)t.cs * a.in0_case;
    int xo[KR];
    unsigned xok = 0;
    evok = 0;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int e = min(64 * k + lane, RV - 1), c = e / C0, ci = e - c * C0, x = t.x0 - 2 + c;
      xok |= (x >= 0 && x < a.W) ? (1u << k) : 0u;
      xo[k] = min(max(x, 0), a.W - 1) * C0 + ci;
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int y = t.y0 - 2 + min(wave + 4 * j, IH - 1);
      const bool yok = y >= 0 && y < a.H;
      const float* rowp = in0 + (int64_t)min(max(y, 0), a.H - 1) * a.P0 * C0;
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        ev[j][k] = rowp[xo[k]];                         // raw: the zero padding is applied when the tile is written, not here
        evok |= (yok && ((xok >> k) & 1u)) ? (1u << (j * KR + k)) : 0u;       // (a select here would wait for the load)
      }
    }
  };
  int it = 0, tile = tile_of(0, total);
  if (tile < 0) return;
  PsmPairTile cur = tile_calc<PSM_PAIR_STEM>(a, tile);
  // one-shot prologue (see psm_pair_up16_kernel): the first tile, conv A's fragments and the biases to registers, conv B's
  // fragments by LDS-DMA -- all requested before anything waits
  issue(cur);
  bf16x8 wA[KS];
  const bf16x8* wa = reinterpret_cast<const bf16x8*>(a.wA) + lane;
#pragma unroll
  for (int s = 0; s < KS; ++s) wA[s] = wa[s * 64];
  const f32x4 bA = *reinterpret_cast<const f32x4*>(a.biasA + 4 * kq), bB = *reinterpret_cast<const f32x4*>(a.biasB + 4 * kq);
  dma_copy(reinterpret_cast<char*>(wbl), a.wB, 6, wave, lane);
  dma_wait_all();                                            // (the first use of `ev` below waits for everything anyway)
#ifdef PSM_STAMPS
  int g_it = 0;
#endif
  while (true) {
    PSTAMP(0);
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
      for (int k = 0; k < KR; ++k)
        img[min(wave + 4 * j, IH - 1) * RV + min(64 * k + lane, RV - 1)] =
            __builtin_bit_cast(unsigned short, (__bf16)(((evok >> (j * KR + k)) & 1u) ? ev[j][k] : 0.f));
    PSTAMP(1);
    lds_barrier();
    PSTAMP(2);
    const bool interior = (cur.flags & 1) != 0;
    const int next = tile_of(++it, total);
    const PsmPairTile nxt = tile_desc(a, next >= 0 ? next : tile);     // past the end: the same tile again (no branch around the loads)
    issue(nxt);
    PSTAMP(3);
    {
      constexpr int R = MH / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = bA;
#pragma unroll
      for (int m = 0; m < R; ++m) {
        if (m == R / 2) __builtin_amdgcn_sched_barrier(0);               // two groups of rows: bounds the gathered operands in flight
        const unsigned short* row = img + (r0 + m) * RV;                 // rows are immediate offsets from the per-lane term offsets
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          unsigned short h[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = row[off[s][j]];
          u32x4 q;
#pragma unroll
          for (int j = 0; j < 4; ++j) q[j] = (unsigned)h[2 * j] | ((unsigned)h[2 * j + 1] << 16);
          acc[m][0] = MFMA_BF(wA[s], __builtin_bit_cast(bf16x8, q), acc[m][0]);
        }
      }
      PSTAMP(4);
      mid_epilogue<1, R, KEEP>(a, mid, cur.cs, cur.y0, cur.x0, r0, xh, lane, interior, acc);
    }
    PSTAMP(5);
    lds_barrier();
    PSTAMP(6);
    {
      constexpr int R = TY / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = bB;
      conv16<1, R, false>(mid, r0, 16 * xh + px, kq, [&](int i) { return wbf[i * 64]; }, acc);
      PSTAMP(7);
      out_epilogue<1, R, true, false>(a, out_tile(a, cur), nullptr, cur.y0, cur.x0, r0, xh, lane, nullptr, acc);
    }
    PSTAMP(8);
#ifdef PSM_STAMPS
    ++g_it;
#endif
    if (next < 0) break;
    tile = next; cur = nxt;
  }
}

// ====================================================================================================================
// LDS-DMA staging (global_load_lds_dwordx4): every lane names its own 16 source bytes, the destination is a wave-uniform LDS
// address + 16 * lane.  The LDS image of a tile ([row][PITCH pixels][PXB bytes]; 64-byte pixels with swizzled slots, the swizzle
// applied on the SOURCE side: a lane fetches the channel group that belongs in its slot) is cut into pieces of 1 KiB, piece i is
// issued by wave i % 4.  A lane's source offset relative to the tile's first source pixel depends on (piece, lane) only, so it is
// computed ONCE per kernel (one VGPR per piece) and a tile costs a wave PW instructions: no staging registers, no LDS stores, no
// address arithmetic per tile.  The tensors are zero-haloed (psm_unet_api.cpp, act_layout), so there is nothing to clamp or to
// zero: the 'same' padding and the overhang of the last tile are read from memory like any other pixel.
//   PXB: bytes per LDS pixel;  PITCH x NROW: the LDS image;  NCOL: source columns that exist for the tile (columns beyond repeat
//   the last one -- they are LDS padding no convolution reads)
// ====================================================================================================================
template <int PXB, int PITCH, int NROW, int NCOL>
struct Dma {
  static constexpr int G = PXB / 16, NI = (NROW * PITCH * G + 63) / 64, PW = (NI + 3) / 4, BYTES = NI * 1024;
  // row_bytes: one source row;  px_bytes: one source pixel;  cb_bytes: first channel of the chunk within the pixel
  static __device__ __forceinline__ void offsets(unsigned (&off)[PW], int wave, int lane, int row_bytes, int px_bytes, int cb_bytes) {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int i = min(wave + 4 * j, NI - 1);                 // a wave without a piece of its own in the last round repeats the last piece
      const int q = i * 64 + lane, P = q / G, sl = q % G;
      const int r = min(P / PITCH, NROW - 1), cl = P % PITCH, c = min(cl, NCOL - 1);
      const int g = PXB == 64 ? ((sl - ((cl >> 1) & 2)) & 3) : sl;      // slot64(): slot sl of column cl holds channel group g
      off[j] = (unsigned)(r * row_bytes + c * px_bytes + cb_bytes + 16 * g);
    }
  }
  static __device__ __forceinline__ void issue(char* tile, const char* src, const unsigned (&off)[PW], int wave) {
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off[j]),
                                       (__attribute__((address_space(3))) void*)(tile + min(wave + 4 * j, NI - 1) * 1024), 16, 0, 0);
  }
};
typedef Dma<64, PL, IH / 2, 17> DmaLow;                    // upsample source at its own resolution, 32 channels
typedef Dma<32, P16, IH, P16> Dma16;                       // 16 channels, same resolution (35 columns: the paired tap of kx = 2 reads one further)
static_assert(DmaLow::BYTES >= LOW_BYTES && Dma16::BYTES >= T16_BYTES, "whole pieces");

// ====================================================================================================================
// level 0, decoder: upsample(32 channels) ++ skip(16 channels) -> 16 -> 16 (+ head).  Weights in LDS, input tiles by LDS-DMA.
// Per tile: [wait for the tile's pieces] barrier | conv A (120 MFMAs) -> mid tile | barrier | request the NEXT tile's pieces
// (every wave is done with the input tiles) | conv B (42 MFMAs) -> stores.  The requests are in flight during conv B, the
// epilogue and the other workgroup of the CU.
// ====================================================================================================================
template <bool KEEP, bool HEAD>
__global__ __launch_bounds__(256, 2) void psm_pair_up16_kernel(PsmPairArgs a) {
  // ONE __shared__ object: with several, hipcc's wait-count pass ties the LDS reads of conv B (mid tile, weights) to the LDS-DMA
  // pieces just requested and drains vmcnt(0) in front of them -- the requests would never be in flight during anything
  // (cdna_hip_programming.md, 'Projection GEMM at M = 256', item 4(a)).  With one object it inserts no such waits at all, so the
  // ordering is this kernel's: dma_wait_all() + barrier before the first read of a DMA'd tile, a barrier after the last.
  constexpr int O_LOW = 0, O_T16 = O_LOW + DmaLow::BYTES, O_MID = O_T16 + Dma16::BYTES, O_W = O_MID + M16_BYTES, O_HEAD = O_W + 21 * 1024;
  __shared__ __attribute__((aligned(16))) char lds[O_HEAD + 272 * 4];
  char* tlow = lds + O_LOW;                                                  // upsample source at its own resolution, 32 channels
  char* t16 = lds + O_T16;                                                   // skip input, 16 channels
  char* mid = lds + O_MID;
  bf16x8* wl = reinterpret_cast<bf16x8*>(lds + O_W);                         // conv A: 9 + 6 fragments, conv B: 6
  float* headw = reinterpret_cast<float*>(lds + O_HEAD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  PSTAMP_ENTRY();
  int it = 0, tile = tile_of(0, total);
  if (tile < 0) return;
  PsmPairTile cur = tile_calc<PSM_PAIR_UPCAT>(a, tile);
  PSTAMP_PRO(0);
  // Everything the workgroup needs before its first MFMA is requested here in ONE go -- biases and head weights to registers, the
  // weight fragments and the first tile by LDS-DMA -- and waited for once, at the top of the tile loop.  (Stamps of the previous
  // form: weights through registers 2.8 us, then biases + head another 0.6 us -- three dependent round trips, 4.9 us from kernel
  // entry to the first MFMA out of a 19 us launch.)
  const f32x4 bA = *reinterpret_cast<const f32x4*>(a.biasA + 4 * kq), bB = *reinterpret_cast<const f32x4*>(a.biasB + 4 * kq);
  float hw0 = 0.f, hw1 = 0.f;
  if (a.head_w) { hw0 = a.head_w[min(tid, 16 * a.head_cout - 1)]; hw1 = a.head_b[min(tid, a.head_cout - 1)]; }
  dma_copy(reinterpret_cast<char*>(wl), a.wA, 15, wave, lane);
  dma_copy(reinterpret_cast<char*>(wl) + 15 * 1024, a.wB, 6, wave, lane);
  unsigned olo[DmaLow::PW], o16[Dma16::PW];
  DmaLow::offsets(olo, wave, lane, a.P0 * 64, 64, 0);
  Dma16::offsets(o16, wave, lane, a.P1 * 32, 32, 0);
  PSTAMP_PRO(1);
  const char* in0 = reinterpret_cast<const char*>(a.in0);
  const char* in1 = reinterpret_cast<const char*>(a.in1);
  DmaLow::issue(tlow, in0 + cur.off0, olo, wave);
  Dma16::issue(t16, in1 + cur.off1, o16, wave);
  PSTAMP_PRO(2);
  zero_mid_pad16(mid, tid);
  if (a.head_w) {
    if (tid < 16 * a.head_cout) headw[tid] = hw0;
    if (tid < a.head_cout) headw[256 + tid] = hw1;
  }
  PSTAMP_PRO(3);
  const bf16x8* wf = wl + lane;
#ifdef PSM_STAMPS
  int g_it = 0;
#endif
  while (true) {
    PSTAMP(0);
    dma_wait_all();
    PSTAMP(1);
    lds_barrier();
    PSTAMP(2);
    const int next = tile_of(++it, total);
    const PsmPairTile nxt = tile_desc(a, next >= 0 ? next : tile);
    {
      constexpr int R = MH / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = bA;
      conv32_up<1, R>(tlow, r0, 16 * xh + px, kq, [&](int i) { return wf[i * 64]; }, acc);
      PSTAMP(3);
      conv16<1, R>(t16, r0, 16 * xh + px, kq, [&](int i) { return wf[(9 + i) * 64]; }, acc);
      PSTAMP(4);
      mid_epilogue<1, R, KEEP>(a, mid, cur.cs, cur.y0, cur.x0, r0, xh, lane, (cur.flags & 1) != 0, acc);
    }
    PSTAMP(5);
    lds_barrier();
    PSTAMP(6);
    if (next >= 0) {                                         // uniform; no register waits for these
      DmaLow::issue(tlow, in0 + nxt.off0, olo, wave);
      Dma16::issue(t16, in1 + nxt.off1, o16, wave);
    }
    {
      constexpr int R = TY / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = bB;
      conv16<1, R>(mid, r0, 16 * xh + px, kq, [&](int i) { return wf[(15 + i) * 64]; }, acc);
      PSTAMP(7);
      out_epilogue<1, R, KEEP || !HEAD, HEAD>(a, (KEEP || !HEAD) ? out_tile(a, cur) : nullptr, HEAD ? head_tile(a, cur) : nullptr, cur.y0, cur.x0,
                                              r0, xh, lane, headw, acc);
    }
    PSTAMP(8);
#ifdef PSM_STAMPS
    ++g_it;
#endif
    if (next < 0) break;
    tile = next; cur = nxt;
  }
}

// ====================================================================================================================
// 32-channel levels: (2x2 max-pool of c0 channels | upsample(c0) ++ skip(c1)) -> 32 -> 32.  Input chunks one after the
// other through one LDS tile, weights through LDS; the mid tile and conv B's weights overlay the staging area.
//   KIND 2: max-pool (c0 a multiple of 16);  KIND 1: upsample ++ skip (c0 a multiple of 32, c1 a multiple of 16)
// ====================================================================================================================
constexpr int W32_BYTES = 18 * 1024;                     // weight fragments of one 32-channel chunk, two channel tiles
constexpr int PAIR32_LDS = T32_BYTES + W32_BYTES;        // 64512 >= M32_BYTES + W32_BYTES
static_assert(P32 % 8 == 0 && PL % 8 == 0, "slot swizzle by column only");
static_assert(M32_BYTES + W32_BYTES <= PAIR32_LDS, "overlay");
static_assert(LOW_BYTES <= T32_BYTES && T16_BYTES <= T32_BYTES, "one staging area for every chunk form");
constexpr int WROUNDS = 5;                               // 18 fragments x 64 pieces / 256 threads, rounded up

// chunk forms of conv A
enum { CK_UP32 = 0, CK_SAME32 = 1, CK_SAME16 = 2, CK_POOL32 = 3, CK_POOL16 = 4, CK_NONE = 5 };
struct Chunk { int form, cb; };                          // form, first channel within its source

template <int KIND, bool KEEP>
__global__ __launch_bounds__(256, 2) void psm_pair32_kernel(PsmPairArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[PAIR32_LDS];
  __shared__ __attribute__((aligned(16))) float bias_l[64];       // conv A's, then conv B's
  char* tile = lds;
  char* wl = lds + T32_BYTES;
  char* mid = lds;                                         // overlays the staging area once conv A is done with it
  char* wbl = lds + M32_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  PSTAMP_ENTRY();
  constexpr int RA = MH / 2, RB = TY / 2;
  const int rA = RA * (wave >> 1), rB = RB * (wave >> 1);
  // conv B's bias: requested here, written to LDS in front of conv B's barrier.  (Written here it was a load -> wait -> ds_write chain at
  // the very top of wave 0, a full round trip before that wave requested anything else: 0.8-1.2 us from entry to the first staging
  // request in the stamps of the one-tile-per-workgroup 128^2 pairs.)
  const float bias_v = (tid < 32 ? a.biasA : a.biasB - 32)[min(tid, 63)];
  const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(wl) + lane;
  auto wget = [&](int i) { return wfrag[i * 64]; };
  // Chunk order: the skip input first (its staging is not prefetched -- at the start of a tile there is nothing to hide it
  // behind), then the upsample source, whose chunks (at its own resolution) are requested during the previous chunk's
  // MFMAs.  Fragment offsets follow pack_pair's order (in0's chunks, then in1's).
  const int n0 = (a.c0 + 31) / 32, n1 = KIND == 1 ? (a.c1 + 31) / 32 : 0;
  const int frag0 = 18 * (a.c0 / 32) + 12 * ((a.c0 % 32) != 0);        // fragments of in0's chunks
  auto chunk_at = [&](int ci) -> Chunk {                    // ci-th chunk in execution order (uniform)
    if (ci < n1) { const int cb = 32 * ci; return {a.c1 - cb >= 32 ? CK_SAME32 : CK_SAME16, cb}; }
    if (ci < n1 + n0) {
      const int cb = 32 * (ci - n1);
      return {KIND == 1 ? CK_UP32 : (a.c0 - cb >= 32 ? CK_POOL32 : CK_POOL16), cb};
    }
    return {CK_NONE, 0};
  };
  auto frag_of = [&](const Chunk& c) {                      // first fragment of the chunk in a.wA
    const bool second = c.form == CK_SAME32 || c.form == CK_SAME16;
    return (second ? frag0 : 0) + 18 * (c.cb / 32);
  };
  auto frags_in = [](int form) { return (form == CK_SAME16 || form == CK_POOL16) ? 12 : 18; };

  f32x4 pf[StLow::NP][1];                                   // prefetched pieces of an upsample-source chunk
  u32x4 pw[WROUNDS];                                        // prefetched weight fragments (a native vector type: an array of HIP's
                                                            // uint4 structs was kept in scratch memory -- every prefetch a round trip)
  unsigned pok = 0;
  for (int it = 0;; ++it) {
    const int tl = tile_of(it, total);
    if (tl < 0) break;
    const PsmPairTile t = tile_calc<KIND == 1 ? PSM_PAIR_UPCAT : PSM_PAIR_POOL>(a, tl);      // (one tile per workgroup where it is planned)
    const int y0 = t.y0, x0 = t.x0;
    const bool interior = (t.flags & 1) != 0;
    const unsigned short* in0 = reinterpret_cast<const unsigned short*>(a.in0) + (int64_t)t.cs * a.in0_case;
    const unsigned short* in1 = reinterpret_cast<const unsigned short*>(a.in1) + (int64_t)t.cs * a.in1_case;
    int tz = tid, lz = lane;                                  // opaque copies, refreshed where they are used: staging positions are
    asm volatile("" : "+v"(tz), "+v"(lz));                    // recomputed there instead of living in registers across the MFMAs
    auto issue_w = [&](const uint4* src, int nfrag) {
#pragma unroll
      for (int u = 0; u < WROUNDS; ++u) pw[u] = reinterpret_cast<const u32x4*>(src)[min(tz + 256 * u, nfrag * 64 - 1)];
    };
    auto write_w = [&](char* dst, int nfrag) {
#pragma unroll
      for (int u = 0; u < WROUNDS; ++u) reinterpret_cast<u32x4*>(dst)[min(tz + 256 * u, nfrag * 64 - 1)] = pw[u];
    };
    f32x4 acc[RA][2];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.biasA + 4 * kq), b1 = *reinterpret_cast<const f32x4*>(a.biasA + 16 + 4 * kq);
#pragma unroll
      for (int m = 0; m < RA; ++m) { acc[m][0] = b0; acc[m][1] = b1; }
    }
#ifdef PSM_STAMPS
    const int g_it = 0;
    int sk = 0;
#define PSTAMP32() do { PSTAMP(sk); ++sk; } while (0)
#else
#define PSTAMP32() do { } while (0)
#endif
    PSTAMP32();
    {
      const Chunk first = chunk_at(0);
      issue_w(a.wA + frag_of(first) * 64, frags_in(first.form));
      if constexpr (KIND == 1)
        if (first.form == CK_UP32) StLow::issue(pf, pok, in0, a.c0, first.cb, a.H / 2, a.W / 2, a.P0, (y0 - 2) / 2, (x0 - 2) / 2, wave, lz, tz);
    }
    // one chunk: stage it (the upsample source: write what was prefetched), request the next chunk's prefetch, MFMAs.  The
    // form is a compile-time tag and every form runs in its own loop below: with one loop over a run-time form the
    // accumulators went through a three-way merge and the MFMAs stopped accumulating in place (twice the registers, spills)
    auto do_chunk = [&](auto form_tag, int cb, int ci) {
      constexpr int FORM = decltype(form_tag)::value;
      lds_barrier();                                         // the previous chunk's (tile's) operands are no longer being read
      asm volatile("" : "+v"(tz), "+v"(lz));
      if constexpr (FORM == CK_UP32) StLow::write(tile, pf, pok, interior, wave, lz, tz);
      else if constexpr (FORM == CK_SAME32) {
        f32x4 v[St32::NP][1]; unsigned ok;
        St32::issue(v, ok, in1, a.c1, cb, a.H, a.W, a.P1, y0 - 2, x0 - 2, wave, lz, tz);
        St32::write(tile, v, ok, interior, wave, lz, tz);
      } else if constexpr (FORM == CK_SAME16) {
        f32x4 v[St16::NP][1]; unsigned ok;
        St16::issue(v, ok, in1, a.c1, cb, a.H, a.W, a.P1, y0 - 2, x0 - 2, wave, lz, tz);
        St16::write(tile, v, ok, interior, wave, lz, tz);
      } else if constexpr (FORM == CK_POOL32) {              // four source pixels per piece: in two halves (registers)
        constexpr int JA = (StPool32::RW + 1) / 2, JB = StPool32::RW - JA;
        { f32x4 v[JA * StPool32::KM][4]; unsigned ok;
          StPool32::template issue<0, JA, false>(v, ok, in0, a.c0, cb, a.H, a.W, a.P0, y0 - 2, x0 - 2, wave, lz, tz);
          StPool32::template write<0, JA, false>(tile, v, ok, interior, wave, lz, tz); }
        { f32x4 v[JB * StPool32::KM + 1][4]; unsigned ok;
          StPool32::template issue<JA, JB, true>(v, ok, in0, a.c0, cb, a.H, a.W, a.P0, y0 - 2, x0 - 2, wave, lz, tz);
          StPool32::template write<JA, JB, true>(tile, v, ok, interior, wave, lz, tz); }
      } else {
        f32x4 v[StPool16::NP][4]; unsigned ok;
        StPool16::issue(v, ok, in0, a.c0, cb, a.H, a.W, a.P0, y0 - 2, x0 - 2, wave, lz, tz);
        StPool16::write(tile, v, ok, interior, wave, lz, tz);
      }
      write_w(wl, frags_in(FORM));
      PSTAMP32();
      lds_barrier();
      PSTAMP32();
      const Chunk nxt = chunk_at(ci + 1);
      asm volatile("" : "+v"(tz), "+v"(lz));
      if constexpr (KIND == 1)
        if (nxt.form == CK_UP32)                             // in flight during this chunk's MFMAs
          StLow::issue(pf, pok, in0, a.c0, nxt.cb, a.H / 2, a.W / 2, a.P0, (y0 - 2) / 2, (x0 - 2) / 2, wave, lz, tz);
      if (nxt.form != CK_NONE) issue_w(a.wA + frag_of(nxt) * 64, frags_in(nxt.form)); else issue_w(a.wB, 18);
      if constexpr (FORM == CK_UP32) conv32_up<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      else if constexpr (FORM == CK_SAME32 || FORM == CK_POOL32) conv32<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      else conv16<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      PSTAMP32();
    };
    {
      int ci = 0;
      if constexpr (KIND == 1) {
        for (int cb = 0; cb + 32 <= a.c1; cb += 32) do_chunk(std::integral_constant<int, CK_SAME32>{}, cb, ci++);
        if (a.c1 & 31) do_chunk(std::integral_constant<int, CK_SAME16>{}, a.c1 & ~31, ci++);
        for (int cb = 0; cb < a.c0; cb += 32) do_chunk(std::integral_constant<int, CK_UP32>{}, cb, ci++);
      } else {
        for (int cb = 0; cb + 32 <= a.c0; cb += 32) do_chunk(std::integral_constant<int, CK_POOL32>{}, cb, ci++);
        if (a.c0 & 31) do_chunk(std::integral_constant<int, CK_POOL16>{}, a.c0 & ~31, ci++);
      }
    }
    lds_barrier();                                           // every wave is done with the staging area: the mid tile goes over it
    asm volatile("" : "+v"(tz), "+v"(lz));
    mid_epilogue<2, RA, KEEP>(a, mid, t.cs, y0, x0, rA, xh, lane, interior, acc);
    zero_mid_pad32(mid, tid);
    write_w(wbl, 18);
    if (tid < 64) bias_l[tid] = bias_v;
    PSTAMP32();
    lds_barrier();
    PSTAMP32();
    {
      f32x4 accb[RB][2];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_l + 32 + 4 * kq), b1 = *reinterpret_cast<const f32x4*>(bias_l + 48 + 4 * kq);
#pragma unroll
        for (int m = 0; m < RB; ++m) { accb[m][0] = b0; accb[m][1] = b1; }
      }
      const bf16x8* wbf = reinterpret_cast<const bf16x8*>(wbl) + lane;
      conv32<2, RB>(mid, rB, 16 * xh + px, kq, [&](int i) { return wbf[i * 64]; }, accb);
      PSTAMP32();
      out_epilogue<2, RB, true, false>(a, out_tile(a, t), nullptr, y0, x0, rB, xh, lane, nullptr, accb);
      PSTAMP32();
    }
  }
}

}  // namespace

// grid: persistent workgroups, two per CU, a multiple of 8 (one share per XCD) when the tile count allows
// Which (kind, channel counts, fused head) combinations the pair kernels below are instantiated for: the ONE predicate the
// planner (psm_unet_api.cpp) and the launcher share.
bool psm_pair_kernel_available(int kind, int cm, int c0, int c1, bool head) {
  if (kind == PSM_PAIR_STEM) return cm == 16 && (c0 == 3 || c0 == 4) && !head;
  if (kind == PSM_PAIR_POOL) return cm == 32 && c0 % 16 == 0 && c0 >= 16 && !head;
  if (kind == PSM_PAIR_UPCAT) {
    if (cm == 16) return c0 == 32 && c1 == 16;                                   // with or without the fused 1x1 head
    if (cm == 32) return c0 % 32 == 0 && c0 >= 32 && c1 % 16 == 0 && c1 >= 16 && !head;
  }
  return false;
}

hipError_t psm_launch_conv_pair(const PsmPairArgs& a, int kind, int cm, int n_cases, hipStream_t st) {
  if (a.tiles_x != (a.W + TX - 1) / TX || a.tiles_y != (a.H + TY - 1) / TY || n_cases < 1 || a.n_cases != n_cases) return hipErrorInvalidValue;
  if (a.head_w && (a.head_cout < 1 || a.head_cout > 16)) return hipErrorInvalidValue;
  const int total = a.tiles_x * a.tiles_y * n_cases;
  static const int wg_max = getenv("PSM_UNET_PAIR_WGS") ? atoi(getenv("PSM_UNET_PAIR_WGS")) : 512;
  int g = total < wg_max ? total : wg_max;                 // (three per CU for the stem kernel, 168 registers: measured, not faster)
  if ((total & 7) == 0 && g >= 8) g &= ~7;
  const dim3 grid((unsigned)g);
  const bool keep = a.mid_out != nullptr, head = a.head_w != nullptr;
  if (!psm_pair_kernel_available(kind, cm, a.c0, a.c1, head)) return hipErrorInvalidValue;
  if (!head && !a.out) return hipErrorInvalidValue;
  if (keep && !a.out) return hipErrorInvalidValue;
#define PAIR_GO(K, ...) do { if (keep) PSM_LAUNCH((K<__VA_ARGS__, true>), grid, dim3(256), 0, st, a); else PSM_LAUNCH((K<__VA_ARGS__, false>), grid, dim3(256), 0, st, a); } while (0)
  if (kind == PSM_PAIR_STEM && cm == 16 && a.c0 == 3 && !head) PAIR_GO(psm_pair_stem16_kernel, 3);
  else if (kind == PSM_PAIR_STEM && cm == 16 && a.c0 == 4 && !head) PAIR_GO(psm_pair_stem16_kernel, 4);
  else if (kind == PSM_PAIR_UPCAT && cm == 16 && a.c0 == 32 && a.c1 == 16) {
    if (keep) { if (head) PSM_LAUNCH((psm_pair_up16_kernel<true, true>), grid, dim3(256), 0, st, a); else PSM_LAUNCH((psm_pair_up16_kernel<true, false>), grid, dim3(256), 0, st, a); }
    else { if (head) PSM_LAUNCH((psm_pair_up16_kernel<false, true>), grid, dim3(256), 0, st, a); else PSM_LAUNCH((psm_pair_up16_kernel<false, false>), grid, dim3(256), 0, st, a); }
  }
  else if (kind == PSM_PAIR_POOL && cm == 32 && a.c0 % 16 == 0 && a.c0 >= 16 && !head) PAIR_GO(psm_pair32_kernel, 2);
  else if (kind == PSM_PAIR_UPCAT && cm == 32 && a.c0 % 32 == 0 && a.c0 >= 32 && a.c1 % 16 == 0 && a.c1 >= 16 && !head) PAIR_GO(psm_pair32_kernel, 1);
  else return hipErrorInvalidValue;
#undef PAIR_GO
  return hipGetLastError();
}
