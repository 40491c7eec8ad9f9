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

// Diagnostic stamps (100 MHz wall clock) of the first, the middle and the last workgroup -- compiled only with -DPSM_STAMPS.
#ifdef PSM_STAMPS
__device__ unsigned long long g_pair_stamps[64];
// slot = 9 * iteration + k of workgroup PSM_STAMP_WG (default 0): the tile loop of one persistent workgroup
#ifndef PSM_STAMP_WG
#define PSM_STAMP_WG 0
#endif
#define PSTAMP(k) do { if (blockIdx.x == PSM_STAMP_WG && threadIdx.x == 0 && 9 * g_it + (k) < 64) g_pair_stamps[9 * g_it + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
hipError_t psm_unet_pair_read_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), sizeof(g_pair_stamps)); }
#else
#define PSTAMP(k) do { } while (0)
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

// diagnostic builds (-DPSM_PAIR_EXP=n, results wrong on purpose): 1 no MFMAs, 2 no tile staging, 3 neither
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 1)
#define MFMA_BF(w, x, c) (c)
#else
#define MFMA_BF(w, x, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((w), (x), (c), 0, 0, 0)
#endif

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the NEXT tile's
// global requests, which are meant to stay in flight across the barriers
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// byte offset of 16-byte channel group `grp` (0..3) of pixel P in a 64-byte-pixel tile (see psm_unet.hip, lds_slot)
__device__ __forceinline__ int slot64(int P, int grp) { return P * 64 + 16 * ((grp + ((P >> 1) & 2)) & 3); }

__device__ __forceinline__ f32x4 bf16x8_max(f32x4 a, f32x4 b) {
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = fmaxf(__uint_as_float(ua[j] << 16), __uint_as_float(ub[j] << 16));
    const float hi = fmaxf(__uint_as_float(ua[j] & 0xffff0000u), __uint_as_float(ub[j] & 0xffff0000u));
    r[j] = (__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16);
  }
  return __builtin_bit_cast(f32x4, r);
}

__device__ __forceinline__ u32x2 pack4(f32x4 v) {
  bf16x4 h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  return __builtin_bit_cast(u32x2, h);
}

// ---- staging: a chunk of CH channels of the (IH x pitch) input tile, from a bf16 NHWC source through MODE -------------
//   MODE 0: same resolution (the skip input)      src [H][W][cpx]
//   MODE 1: 2x nearest-neighbour upsample         src [H/2][W/2][cpx]
//   MODE 2: 2x2 max-pool                          src [2H][2W][cpx]
// tile_issue requests rounds [B0, B0 + NR) of the chunk (clamped addresses, nothing depends on the data), tile_write
// combines (2x2 max), zeroes what lies outside the image and stores to LDS.  The persistent kernels issue the NEXT tile's
// requests before the current tile's MFMAs and write them when the current tile no longer needs its LDS buffers.
template <int CH> struct TileGeom {
  static constexpr int PITCH = CH == 32 ? P32 : P16;             // LDS pitch
  static constexpr int COLS = CH == 32 ? 34 : P16;               // staged columns
  static constexpr int G = CH / 8;
  static constexpr int N = COLS * IH * G;
  static constexpr int ROUNDS = (N + 255) / 256;                 // 10 (32 channels) / 5 (16 channels)
};
// the upsample source at its own resolution: (IH/2) x (P32/2) low-resolution pixels of 32 channels (tile origins are even)
constexpr int LOW_COLS = 17;                                                             // (34 + 0) / 2 low-resolution columns
constexpr int LOW_N = (IH / 2) * LOW_COLS * 4, LOW_ROUNDS = (LOW_N + 255) / 256;      // 612 pieces, 3 rounds
constexpr int PL = 24;                                                                   // LDS pitch of the low-resolution tile (17 used)
constexpr int LOW_BYTES = (IH / 2) * PL * 64;                                            // 13824
template <int NV>
__device__ __forceinline__ void low_issue(f32x4 (&v)[NV][1], unsigned& okmask, const unsigned short* src, int cpx, int cb,
                                          int H, int W, int y0, int x0, int tid) {
  const int Hs = H / 2, Ws = W / 2, yl = (y0 - 2) / 2, xl = (x0 - 2) / 2;     // y0, x0 even: exact (also for -2)
  okmask = 0;
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
  return;
#endif
#pragma unroll
  for (int u = 0; u < LOW_ROUNDS; ++u) {
    const int q = min(tid + 256 * u, LOW_N - 1);
    const int pos = q >> 2, g = q & 3;
    const int r = pos / LOW_COLS, c = pos - r * LOW_COLS;
    const int y = yl + r, x = xl + c;
    okmask |= (y >= 0 && y < Hs && x >= 0 && x < Ws) ? (1u << u) : 0u;
    v[u][0] = *reinterpret_cast<const f32x4*>(src + ((int64_t)min(max(y, 0), Hs - 1) * Ws + min(max(x, 0), Ws - 1)) * cpx + cb + 8 * g);
  }
}
template <int NV>
__device__ __forceinline__ void low_write(char* tile, const f32x4 (&v)[NV][1], unsigned okmask, int tid) {
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
  return;
#endif
#pragma unroll
  for (int u = 0; u < LOW_ROUNDS; ++u) {
    const int q = min(tid + 256 * u, LOW_N - 1);
    f32x4 t = v[u][0];
    const bool ok = (okmask >> u) & 1u;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = ok ? t[j] : 0.f;
    *reinterpret_cast<f32x4*>(tile + slot64(((q >> 2) / LOW_COLS) * PL + (q >> 2) % LOW_COLS, q & 3)) = t;
  }
}
template <int MODE, int CH, int B0, int NR, int NV>
__device__ __forceinline__ void tile_issue(f32x4 (&v)[NV][MODE == 2 ? 4 : 1], unsigned& okmask, const unsigned short* src, int cpx, int cb,
                                           int H, int W, int y0, int x0, int tid) {
  using T = TileGeom<CH>;
  const int Ws = MODE == 1 ? W / 2 : (MODE == 2 ? 2 * W : W);
  okmask = 0;
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
  return;
#endif
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int q = min(tid + 256 * (B0 + u), T::N - 1);          // surplus threads repeat the last piece (same value, same place)
    const int pos = q / T::G, g = q - pos * T::G;
    const int r = pos / T::COLS, c = pos - r * T::COLS;
    const int y = y0 - 2 + r, x = x0 - 2 + c;
    okmask |= (y >= 0 && y < H && x >= 0 && x < W) ? (1u << u) : 0u;
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    if (MODE == 2) {
      const unsigned short* p = src + ((int64_t)(2 * yc) * Ws + 2 * xc) * cpx + cb + 8 * g;
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][MODE == 2 ? 1 : 0] = *reinterpret_cast<const f32x4*>(p + cpx);
      v[u][MODE == 2 ? 2 : 0] = *reinterpret_cast<const f32x4*>(p + (int64_t)Ws * cpx);
      v[u][MODE == 2 ? 3 : 0] = *reinterpret_cast<const f32x4*>(p + (int64_t)Ws * cpx + cpx);
    } else if (MODE == 1) {
      v[u][0] = *reinterpret_cast<const f32x4*>(src + ((int64_t)(yc >> 1) * Ws + (xc >> 1)) * cpx + cb + 8 * g);
    } else {
      v[u][0] = *reinterpret_cast<const f32x4*>(src + ((int64_t)yc * Ws + xc) * cpx + cb + 8 * g);
    }
  }
}
template <int MODE, int CH, int B0, int NR, int NV>
__device__ __forceinline__ void tile_write(char* tile, const f32x4 (&v)[NV][MODE == 2 ? 4 : 1], unsigned okmask, int tid) {
  using T = TileGeom<CH>;
#if defined(PSM_PAIR_EXP) && (PSM_PAIR_EXP & 2)
  return;
#endif
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int q = min(tid + 256 * (B0 + u), T::N - 1);
    const int pc = q / T::G, g = q - pc * T::G;
    const int pos = (pc / T::COLS) * T::PITCH + pc % T::COLS;
    f32x4 t = v[u][0];
    if (MODE == 2) t = bf16x8_max(bf16x8_max(v[u][0], v[u][MODE == 2 ? 1 : 0]), bf16x8_max(v[u][MODE == 2 ? 2 : 0], v[u][MODE == 2 ? 3 : 0]));
    const bool ok = (okmask >> u) & 1u;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = ok ? t[j] : 0.f;
    *reinterpret_cast<f32x4*>(tile + (CH == 32 ? slot64(pos, g) : pos * 32 + g * 16)) = t;
  }
}
// request + write of a whole chunk, five rounds at a time (register budget of the chunked kernel)
template <int MODE, int CH>
__device__ __forceinline__ void stage_tile(char* tile, const unsigned short* src, int cpx, int cb, int H, int W, int y0, int x0, int tid) {
  f32x4 v[5][MODE == 2 ? 4 : 1];
  unsigned ok;
  tile_issue<MODE, CH, 0, 5>(v, ok, src, cpx, cb, H, W, y0, x0, tid);
  tile_write<MODE, CH, 0, 5>(tile, v, ok, tid);
  if constexpr (TileGeom<CH>::ROUNDS > 5) {
    tile_issue<MODE, CH, 5, 5>(v, ok, src, cpx, cb, H, W, y0, x0, tid);
    tile_write<MODE, CH, 5, 5>(tile, v, ok, tid);
  }
}

// ---- one chunk of a convolution on an LDS tile --------------------------------------------------------------------
// acc[m][nt]: row r0 + m, channel tile nt of the wave's 16-pixel half.  A chunk is S steps (tap columns); per step the
// NX input rows and the 3 NT weight fragments are read from LDS one step ahead of the MFMAs that use them (two register
// stages; nothing moves across the sched_barriers, so the live set stays at two stages).
//   xload(s, i): operand of step s, input-row slot i;   wload(s, ky, nt): weight fragment;   HALF: slot of row j is j >> 1
template <int NT, int R, int NX, int S, bool HALF, bool DB, typename XLoad, typename WLoad>
__device__ __forceinline__ void conv_steps(XLoad xload, WLoad wload, f32x4 (&acc)[R][NT]) {
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
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < R; ++m)
          acc[m][nt] = MFMA_BF(wb[DB ? (s & 1) : 0][ky * NT + nt], xb[DB ? (s & 1) : 0][HALF ? (m + ky) >> 1 : m + ky], acc[m][nt]);
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
template <int NT, int R, typename WGet>
__device__ __forceinline__ void conv32_up(const char* tile, int r0, int xcol, int kq, WGet wget, f32x4 (&acc)[R][NT]) {
  static_assert((R & 1) == 0, "row pairs");
  conv_steps<NT, R, (R + 2) / 2, 3, true, NT == 1>(
      [&](int kx, int i) { return *reinterpret_cast<const bf16x8*>(tile + (r0 >> 1) * (PL * 64) + slot64((xcol + kx) >> 1, kq) + i * (PL * 64)); },
      [&](int kx, int ky, int nt) { return wget((kx * 3 + ky) * NT + nt); }, acc);
}
// 16-channel chunk, 32-byte pixels, pitch P16: step = tap pair (kx = 2s, 2s + 1); fragments [s][ky][nt]
template <int NT, int R, typename WGet>
__device__ __forceinline__ void conv16(const char* tile, int r0, int xcol, int kq, WGet wget, f32x4 (&acc)[R][NT]) {
  conv_steps<NT, R, R + 2, 2, false, NT == 1>(
      [&](int s, int i) { return *reinterpret_cast<const bf16x8*>(tile + ((r0 + i) * P16 + xcol + 2 * s) * 32 + kq * 16); },
      [&](int s, int ky, int nt) { return wget((s * 3 + ky) * NT + nt); }, acc);
}

// ---- conv A's epilogue: bias + ReLU, zero outside the image, bf16 into the mid tile (LDS) --------------------------
// b[nt]: the lane's four bias values (channels 16 nt + 4 (lane >> 4) ..), loaded at kernel start
// KEEP (introspection builds of the launch): the activation is stored to HBM as well, every pixel by the tile that owns it
template <int NT, int R, bool KEEP>
__device__ __forceinline__ void mid_epilogue(const PsmPairArgs& a, char* mid, int cs, int y0, int x0, int r0, int xh, int lane,
                                             const f32x4 (&b)[NT], f32x4 (&acc)[R][NT]) {
  const int px = lane & 15, kq = lane >> 4;
  const int col = 16 * xh + px, x = x0 - 1 + col;
  const bool xok = x >= 0 && x < a.W;
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const int mr = r0 + m, y = y0 - 1 + mr;
    const bool ok = xok && y >= 0 && y < a.H;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = acc[m][nt] + b[nt];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ok ? fmaxf(v[j], 0.f) : 0.f;
      const u32x2 h = pack4(v);
      if (NT == 1) *reinterpret_cast<u32x2*>(mid + (mr * P16 + col) * 32 + kq * 8) = h;
      else *reinterpret_cast<u32x2*>(mid + slot64(mr * P32 + col, 2 * nt + (kq >> 1)) + (kq & 1) * 8) = h;
      if (KEEP && ok && mr >= 1 && mr <= TY && col >= 1 && col <= TX)
        *reinterpret_cast<u32x2*>(a.mid_out + (int64_t)cs * a.out_case + ((int64_t)y * a.W + x) * (16 * NT) + 16 * nt + 4 * kq) = h;
    }
  }
}

// ---- conv B's epilogue: bias + ReLU, bf16 NHWC store (STORE), fused linear 1x1 head (HEAD) --------------------------
// headw (LDS): [16][head_cout] weights, then head_cout biases at [256].  The head sums a pixel's 16 channels: four in the
// lane, then across the four lanes l, l ^ 16, l ^ 32, l ^ 48 with v_permlane32_swap / v_permlane16_swap (no LDS round trip)
__device__ __forceinline__ float sum_lane_groups(float s) {
  typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
  const u32x2v p = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  const float t = __uint_as_float(p[0]) + __uint_as_float(p[1]);
  const u32x2v q = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
template <int NT, int R, bool STORE, bool HEAD>
__device__ __forceinline__ void out_epilogue(const PsmPairArgs& a, int cs, int y0, int x0, int r0, int xh, int lane,
                                             const f32x4 (&b)[NT], const float* headw, f32x4 (&acc)[R][NT]) {
  static_assert(!HEAD || NT == 1, "the fused head reads one channel tile");
  const int px = lane & 15, kq = lane >> 4;
  const int col = 16 * xh + px, x = x0 + col;
  const bool xok = col < TX && x < a.W;
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const int y = y0 + r0 + m;
    const bool ok = xok && y < a.H;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v = acc[m][nt] + b[nt];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      acc[m][nt] = v;
      if (STORE && ok)
        *reinterpret_cast<u32x2*>(a.out + (int64_t)cs * a.out_case + ((int64_t)y * a.W + x) * (16 * NT) + 16 * nt + 4 * kq) = pack4(v);
    }
  }
  if constexpr (HEAD) {
    for (int o = 0; o < a.head_cout; ++o) {
      f32x4 hw;
#pragma unroll
      for (int j = 0; j < 4; ++j) hw[j] = headw[(4 * kq + j) * a.head_cout + o];
      const float hb = headw[256 + o];
      float s[R];
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const f32x4 v = acc[m][0];
        s[m] = sum_lane_groups(v[0] * hw[0] + v[1] * hw[1] + v[2] * hw[2] + v[3] * hw[3]) + hb;
      }
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int y = y0 + r0 + m;
        if (kq == 0 && xok && y < a.H) a.head_out[(int64_t)cs * a.head_case + ((int64_t)y * a.W + x) * a.head_cout + o] = s[m];
      }
    }
  }
}
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
struct TilePos { int cs, y0, x0; };
__device__ __forceinline__ TilePos tile_pos(const PsmPairArgs& a, int t) {
  const int per_case = a.tiles_x * a.tiles_y;
  const int cs = t / per_case, r = t - cs * per_case, by = r / a.tiles_x;
  return {cs, by * TY, (r - by * a.tiles_x) * TX};
}

// ====================================================================================================================
// level 0, encoder: raw image (C0 = 3 or 4 float32 channels) -> 16 -> 16.  conv A flattens k = tap*C0 + channel and
// pads it to KS steps of 32; its operand is gathered from the bf16 image tile with per-lane fixed offsets.
// ====================================================================================================================
template <int C0, bool KEEP>
__global__ __launch_bounds__(256, 2) void psm_pair_stem16_kernel(PsmPairArgs a) {
  constexpr int K = 9 * C0, KS = (K + 31) / 32;
  constexpr int PI = 34;                                  // pitch of the image tile (pixels)
  constexpr int NI = IH * PI * C0;                       // image tile values (bf16), pitch 34
  constexpr int ROUNDS = (NI + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned short img[ROUNDS * 256];
  __shared__ __attribute__((aligned(16))) char mid[M16_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  bf16x8 wA[KS], wB[6];
  const bf16x8* wa = reinterpret_cast<const bf16x8*>(a.wA) + lane;
  const bf16x8* wb = reinterpret_cast<const bf16x8*>(a.wB) + lane;
#pragma unroll
  for (int s = 0; s < KS; ++s) wA[s] = wa[s * 64];
#pragma unroll
  for (int s = 0; s < 6; ++s) wB[s] = wb[s * 64];
  const f32x4 bA[1] = {*reinterpret_cast<const f32x4*>(a.biasA + 4 * kq)}, bB[1] = {*reinterpret_cast<const f32x4*>(a.biasB + 4 * kq)};
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
  float ev[ROUNDS];
  auto issue = [&](const TilePos& t, int tz) {
    const float* in0 = reinterpret_cast<const float*>(a.in0) + (int64_t)t.cs * a.in0_case;
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
      const int e = min(tz + 256 * u, NI - 1), pos = e / C0, ci = e - pos * C0;
      const int r = pos / PI, c = pos - r * PI;
      const int y = t.y0 - 2 + r, x = t.x0 - 2 + c;
      const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
      const float v = in0[((int64_t)min(max(y, 0), a.H - 1) * a.W + min(max(x, 0), a.W - 1)) * C0 + ci];
      ev[u] = ok ? v : 0.f;
    }
  };
  int it = 0, tile = tile_of(0, total);
  if (tile < 0) return;
  TilePos cur = tile_pos(a, tile);
  issue(cur, tid);
  while (true) {
    int tz = tid;                                                      // opaque copy: staging positions recomputed per tile, not kept in registers
    asm volatile("" : "+v"(tz));
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) img[tz + 256 * u] = __builtin_bit_cast(unsigned short, (__bf16)ev[u]);
    lds_barrier();
    const int next = tile_of(++it, total);
    const TilePos nxt = tile_pos(a, next >= 0 ? next : tile);          // past the end: the same tile again (no branch around the loads)
    issue(nxt, tz);
    {
      constexpr int R = MH / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const unsigned short* row = img + (r0 + m) * (PI * C0);          // rows are immediate offsets from the per-lane term offsets
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
      mid_epilogue<1, R, KEEP>(a, mid, cur.cs, cur.y0, cur.x0, r0, xh, lane, bA, acc);
    }
    lds_barrier();
    {
      constexpr int R = TY / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      conv16<1, R>(mid, r0, 16 * xh + px, kq, [&](int i) { return wB[i]; }, acc);
      out_epilogue<1, R, true, false>(a, cur.cs, cur.y0, cur.x0, r0, xh, lane, bB, nullptr, acc);
    }
    if (next < 0) break;
    tile = next; cur = nxt;
  }
}

// ====================================================================================================================
// level 0, decoder: upsample(32 channels) ++ skip(16 channels) -> 16 -> 16 (+ head).  All weights in registers.
// ====================================================================================================================
template <bool KEEP, bool HEAD>
__global__ __launch_bounds__(256, 2) void psm_pair_up16_kernel(PsmPairArgs a) {
  __shared__ __attribute__((aligned(16))) char tlow[LOW_BYTES];             // upsample source at its own resolution, 32 channels
  __shared__ __attribute__((aligned(16))) char t16[T16_BYTES];              // skip input, 16 channels
  __shared__ __attribute__((aligned(16))) char mid[M16_BYTES];
  __shared__ __attribute__((aligned(16))) bf16x8 wl[21 * 64];               // conv A: 9 + 6 fragments, conv B: 6
  __shared__ float headw[272];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  for (int i = tid; i < 21 * 64; i += 256)
    reinterpret_cast<uint4*>(wl)[i] = i < 15 * 64 ? a.wA[i] : a.wB[i - 15 * 64];
  const f32x4 bA[1] = {*reinterpret_cast<const f32x4*>(a.biasA + 4 * kq)}, bB[1] = {*reinterpret_cast<const f32x4*>(a.biasB + 4 * kq)};
  load_head(a, headw, tid);
  zero_mid_pad16(mid, tid);
  const bf16x8* wf = wl + lane;
  f32x4 vlo[LOW_ROUNDS][1], v16[5][1];
  unsigned oklo, ok16;
  auto issue = [&](const TilePos& t, int tz) {
    low_issue(vlo, oklo, reinterpret_cast<const unsigned short*>(a.in0) + (int64_t)t.cs * a.in0_case, 32, 0, a.H, a.W, t.y0, t.x0, tz);
    tile_issue<0, 16, 0, 5>(v16, ok16, reinterpret_cast<const unsigned short*>(a.in1) + (int64_t)t.cs * a.in1_case, 16, 0, a.H, a.W, t.y0, t.x0, tz);
  };
  int it = 0, tile = tile_of(0, total);
  if (tile < 0) return;
  TilePos cur = tile_pos(a, tile);
  issue(cur, tid);
#ifdef PSM_STAMPS
  int g_it = 0;
#endif
  while (true) {
    // the per-thread staging positions are recomputed per tile from an opaque copy of tid: hoisted out of the loop they
    // would sit in ~40 registers for the whole kernel
    int tz = tid;
    asm volatile("" : "+v"(tz));
    PSTAMP(0);
    low_write(tlow, vlo, oklo, tz);
    tile_write<0, 16, 0, 5>(t16, v16, ok16, tz);
    PSTAMP(1);
    lds_barrier();
    PSTAMP(2);
    const int next = tile_of(++it, total);
    const TilePos nxt = tile_pos(a, next >= 0 ? next : tile);          // past the end: the same tile again (no branch around the loads)
    issue(nxt, tz);
    {
      constexpr int R = MH / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      conv32_up<1, R>(tlow, r0, 16 * xh + px, kq, [&](int i) { return wf[i * 64]; }, acc);
      PSTAMP(3);
      conv16<1, R>(t16, r0, 16 * xh + px, kq, [&](int i) { return wf[(9 + i) * 64]; }, acc);
      PSTAMP(4);
      mid_epilogue<1, R, KEEP>(a, mid, cur.cs, cur.y0, cur.x0, r0, xh, lane, bA, acc);
    }
    PSTAMP(5);
    lds_barrier();
    PSTAMP(6);
    {
      constexpr int R = TY / 2;
      const int r0 = R * (wave >> 1);
      f32x4 acc[R][1];
#pragma unroll
      for (int m = 0; m < R; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      conv16<1, R>(mid, r0, 16 * xh + px, kq, [&](int i) { return wf[(15 + i) * 64]; }, acc);
      PSTAMP(7);
      out_epilogue<1, R, KEEP || !HEAD, HEAD>(a, cur.cs, cur.y0, cur.x0, r0, xh, lane, bB, headw, acc);
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
  char* tile = lds;
  char* wl = lds + T32_BYTES;
  char* mid = lds;                                         // overlays the staging area once conv A is done with it
  char* wbl = lds + M32_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kq = lane >> 4, xh = wave & 1;
  const int total = a.tiles_x * a.tiles_y * a.n_cases;
  constexpr int RA = MH / 2, RB = TY / 2;
  const int rA = RA * (wave >> 1), rB = RB * (wave >> 1);
  const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(wl) + lane;
  auto wget = [&](int i) { return wfrag[i * 64]; };
  // Chunk order: the skip input first (its ten staging rounds are not prefetched -- at the start of a tile there is nothing
  // to hide them behind), then the upsample source, whose chunks (three rounds at its own resolution) are requested during
  // the previous chunk's MFMAs.  Fragment offsets follow pack_pair's order (in0's chunks, then in1's).
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

  f32x4 pf[LOW_ROUNDS][1];                                  // prefetched pieces of an upsample-source chunk
  uint4 pw[WROUNDS];                                        // prefetched weight fragments
  unsigned pok = 0;
  for (int it = 0;; ++it) {
    const int tl = tile_of(it, total);
    if (tl < 0) break;
    const TilePos t = tile_pos(a, tl);
    const int y0 = t.y0, x0 = t.x0;
    const unsigned short* in0 = reinterpret_cast<const unsigned short*>(a.in0) + (int64_t)t.cs * a.in0_case;
    const unsigned short* in1 = reinterpret_cast<const unsigned short*>(a.in1) + (int64_t)t.cs * a.in1_case;
    int tz = tid;                                             // opaque copy, refreshed where it is used: the staging positions are
    asm volatile("" : "+v"(tz));                              // recomputed there instead of living in registers across the MFMAs
    auto issue_w = [&](const uint4* src, int nfrag) {
#pragma unroll
      for (int u = 0; u < WROUNDS; ++u) pw[u] = src[min(tz + 256 * u, nfrag * 64 - 1)];
    };
    auto write_w = [&](char* dst, int nfrag) {
#pragma unroll
      for (int u = 0; u < WROUNDS; ++u) reinterpret_cast<uint4*>(dst)[min(tz + 256 * u, nfrag * 64 - 1)] = pw[u];
    };
    f32x4 acc[RA][2];
#pragma unroll
    for (int m = 0; m < RA; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#ifdef PSM_STAMPS
    const int g_it = 0;
    int sk = 0;
#define PSTAMP32() do { PSTAMP(sk); ++sk; } while (0)
#else
#define PSTAMP32() do { } while (0)
#endif
    PSTAMP32();
    Chunk cur = chunk_at(0);
    issue_w(a.wA + frag_of(cur) * 64, frags_in(cur.form));
    if (cur.form == CK_UP32) low_issue(pf, pok, in0, a.c0, cur.cb, a.H, a.W, y0, x0, tz);
    for (int ci = 0; cur.form != CK_NONE; ++ci) {
      lds_barrier();                                         // the previous chunk's (tile's) operands are no longer being read
      asm volatile("" : "+v"(tz));
      if (cur.form == CK_UP32) low_write(tile, pf, pok, tz);
      else if (cur.form == CK_SAME32) stage_tile<0, 32>(tile, in1, a.c1, cur.cb, a.H, a.W, y0, x0, tz);
      else if (cur.form == CK_SAME16) stage_tile<0, 16>(tile, in1, a.c1, cur.cb, a.H, a.W, y0, x0, tz);
      else if (cur.form == CK_POOL32) stage_tile<2, 32>(tile, in0, a.c0, cur.cb, a.H, a.W, y0, x0, tz);
      else stage_tile<2, 16>(tile, in0, a.c0, cur.cb, a.H, a.W, y0, x0, tz);
      write_w(wl, frags_in(cur.form));
      PSTAMP32();
      lds_barrier();
      PSTAMP32();
      const Chunk nxt = chunk_at(ci + 1);
      asm volatile("" : "+v"(tz));
      if (nxt.form == CK_UP32) low_issue(pf, pok, in0, a.c0, nxt.cb, a.H, a.W, y0, x0, tz);      // in flight during this chunk's MFMAs
      if (nxt.form != CK_NONE) issue_w(a.wA + frag_of(nxt) * 64, frags_in(nxt.form)); else issue_w(a.wB, 18);
      if (cur.form == CK_UP32) conv32_up<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      else if (cur.form == CK_SAME32 || cur.form == CK_POOL32) conv32<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      else conv16<2, RA>(tile, rA, 16 * xh + px, kq, wget, acc);
      PSTAMP32();
      cur = nxt;
    }
    f32x4 bA[2], bB[2];                                      // requested here: they land during the barrier and the first epilogue rows
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      bA[nt] = *reinterpret_cast<const f32x4*>(a.biasA + 16 * nt + 4 * kq);
      bB[nt] = *reinterpret_cast<const f32x4*>(a.biasB + 16 * nt + 4 * kq);
    }
    lds_barrier();                                           // every wave is done with the staging area: the mid tile goes over it
    asm volatile("" : "+v"(tz));
    mid_epilogue<2, RA, KEEP>(a, mid, t.cs, y0, x0, rA, xh, lane, bA, acc);
    zero_mid_pad32(mid, tid);
    write_w(wbl, 18);
    PSTAMP32();
    lds_barrier();
    PSTAMP32();
    {
      f32x4 accb[RB][2];
#pragma unroll
      for (int m = 0; m < RB; ++m) { accb[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accb[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      const bf16x8* wbf = reinterpret_cast<const bf16x8*>(wbl) + lane;
      conv32<2, RB>(mid, rB, 16 * xh + px, kq, [&](int i) { return wbf[i * 64]; }, accb);
      PSTAMP32();
      out_epilogue<2, RB, true, false>(a, t.cs, y0, x0, rB, xh, lane, bB, nullptr, accb);
      PSTAMP32();
    }
  }
}

}  // namespace

// grid: persistent workgroups, two per CU, a multiple of 8 (one share per XCD) when the tile count allows
hipError_t psm_launch_conv_pair(const PsmPairArgs& a, int kind, int cm, int n_cases, hipStream_t st) {
  if (a.tiles_x != (a.W + TX - 1) / TX || a.tiles_y != (a.H + TY - 1) / TY || n_cases < 1 || a.n_cases != n_cases) return hipErrorInvalidValue;
  if (a.head_w && (a.head_cout < 1 || a.head_cout > 16)) return hipErrorInvalidValue;
  const int total = a.tiles_x * a.tiles_y * n_cases;
  static const int wg_max = getenv("PSM_UNET_PAIR_WGS") ? atoi(getenv("PSM_UNET_PAIR_WGS")) : 512;
  int g = total < wg_max ? total : wg_max;
  if ((total & 7) == 0 && g >= 8) g &= ~7;
  const dim3 grid((unsigned)g);
  const bool keep = a.mid_out != nullptr, head = a.head_w != nullptr;
  if (!head && !a.out) return hipErrorInvalidValue;
  if (keep && !a.out) return hipErrorInvalidValue;
#define PAIR_GO(K, ...) do { if (keep) hipLaunchKernelGGL((K<__VA_ARGS__, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((K<__VA_ARGS__, false>), grid, dim3(256), 0, st, a); } while (0)
  if (kind == PSM_PAIR_STEM && cm == 16 && a.c0 == 3 && !head) PAIR_GO(psm_pair_stem16_kernel, 3);
  else if (kind == PSM_PAIR_STEM && cm == 16 && a.c0 == 4 && !head) PAIR_GO(psm_pair_stem16_kernel, 4);
  else if (kind == PSM_PAIR_UPCAT && cm == 16 && a.c0 == 32 && a.c1 == 16) {
    if (keep) { if (head) hipLaunchKernelGGL((psm_pair_up16_kernel<true, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((psm_pair_up16_kernel<true, false>), grid, dim3(256), 0, st, a); }
    else { if (head) hipLaunchKernelGGL((psm_pair_up16_kernel<false, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((psm_pair_up16_kernel<false, false>), grid, dim3(256), 0, st, a); }
  }
  else if (kind == PSM_PAIR_POOL && cm == 32 && a.c0 % 16 == 0 && a.c0 >= 16 && !head) PAIR_GO(psm_pair32_kernel, 2);
  else if (kind == PSM_PAIR_UPCAT && cm == 32 && a.c0 % 32 == 0 && a.c0 >= 32 && a.c1 % 16 == 0 && a.c1 >= 16 && !head) PAIR_GO(psm_pair32_kernel, 1);
  else return hipErrorInvalidValue;
#undef PAIR_GO
  return hipGetLastError();
}
