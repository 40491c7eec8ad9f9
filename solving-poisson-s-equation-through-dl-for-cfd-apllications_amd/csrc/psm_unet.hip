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

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef PSM_CONV_PD
#define PSM_CONV_PD 1
#endif
// bf16 form: activation rows kept across ky (round 4 experiment).  Measured A/B on one box, 8 cases per step: 157.3 / 157.2 us
// with it, 154.4 / 155.1 us without (profiles/archive/r04_conv_experiments.txt) -- a third of the LDS operand reads gone and the pass
// 1.5 % SLOWER: the matrix phase is not what these layers wait for.  Off; -DPSM_CONV_KYREUSE=1 builds it.
#ifndef PSM_CONV_KYREUSE
#define PSM_CONV_KYREUSE 0
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// activation stores: -DPSM_NT_ACT streams them past the L2 (the consumer is the next launch, on any XCD)
#ifdef PSM_NT_ACT
#define PSM_ACT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define PSM_ACT_STORE(p, v) (*(p) = (v))
#endif

// Diagnostic stamps (100 MHz wall clock) of workgroup (0,0,0) -- compiled only with -DPSM_STAMPS.
#ifdef PSM_STAMPS
__device__ unsigned long long g_unet_stamps[64];
#define USTAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (k) < 64) g_unet_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
hipError_t psm_unet_read_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_unet_stamps), sizeof(g_unet_stamps)); }
#else
#define USTAMP(k) do { } while (0)
hipError_t psm_unet_read_stamps(unsigned long long* out) { for (int i = 0; i < 64; ++i) out[i] = 0; return hipSuccess; }
#endif

#ifndef PSM_WT_STORES
#define PSM_WT_STORES 1
#endif

namespace {

// fused linear 1x1 head on the finished 16-channel tiles of a wave: v[m][r] = activation of (row m, pixel lane & 15,
// channel 4 * (lane >> 4) + r).  In-lane part of the 16-channel sum, then the four lanes l, l ^ 16, l ^ 32, l ^ 48 of a pixel.
// The head's weight is loaded ONCE per output channel and only after every activation store has been issued: a load between
// the stores would make each of them a full round trip (stores count in vmcnt too).
template <int WM>
__device__ __forceinline__ void head_epilogue(const PsmConvArgs& a, int cs, int y_first, int x0, int lane, const f32x4 (&v)[WM]) {
  const int px = lane & 15, kq = lane >> 4;
  for (int o = 0; o < a.head_cout; ++o) {
    f32x4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = a.head_w[(4 * kq + r) * a.head_cout + o];
    const float hb = a.head_b[o];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int y = y_first + m;
      float s = v[m][0] * w[0] + v[m][1] * w[1] + v[m][2] * w[2] + v[m][3] * w[3];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (kq == 0 && y < a.H && x0 + px < a.W)
        a.head_out[(int64_t)cs * a.head_case + ((int64_t)y * a.W + x0 + px) * a.head_cout + o] = s + hb;
    }
  }
}

constexpr int TW = 16;            // tile width (pixels) = MFMA rows
constexpr int CC = 16;            // input channels per chunk
constexpr int LDC = 16;           // LDS pixel stride (floats): 64 B = four 16-byte slots, no padding
// 16-byte slot (word offset) of channel group `grp` (0..3) of LDS pixel P.  A ds_read_b128 is served in four groups of 16
// lanes that are NOT contiguous -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): each holds
// all 16 pixels of an MFMA row, half of them with channel group kq and half with kq+1.  Rotating the slots by two for
// pixels with bit 2 set makes every such group hit 16 distinct slots of the 256-byte bank row, for any tile row and
// tap shift (a padded 80-byte stride gave 2-way conflicts on 3 of 16 lanes: 40 % of the LDS cycles).
__device__ __forceinline__ int lds_slot(int P, int grp) { return P * LDC + 4 * ((grp + ((P >> 1) & 2)) & 3); }

// One finished group of four channels.  Straight-line on purpose (clamped addresses, selects, fully unrolled
// slab sums): every load of a chunk's staging must be in flight together -- a load inside a branch or a
// runtime loop costs a drained vmcnt, i.e. one full memory round trip each.
//   KSM == 1: the producer stored finished activations;  KSM == 8: it may have stored up to 8 partial-sum
//   slabs (split-K): they are added in slab order, then the producer's bias and ReLU.
template <int KSM>
__device__ __forceinline__ f32x4 read4(const float* p, int ks, int64_t slab, const float* pbias, int ch) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  if (KSM > 1) {
#pragma unroll
    for (int s = 1; s < KSM; ++s) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + (int64_t)min(s, ks - 1) * slab);
      const float keep = s < ks ? 1.f : 0.f;
      v += t * keep;
    }
    const f32x4 b = *reinterpret_cast<const f32x4*>(pbias + ch);
    const bool fin = ks > 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fin ? fmaxf(v[j] + b[j], 0.f) : v[j];
  }
  return v;
}

// four consecutive channels [ch, ch+4) of the concatenated input at output-resolution pixel (y, x);
// channel counts are multiples of 4 here, so a group never straddles the concatenation seam
template <int SRC, int KSM>
__device__ __forceinline__ f32x4 fetch4(const PsmConvArgs& a, const float* in0, const float* in1, int y, int x, int ch) {
  const bool inside = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
  const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
  const bool first = ch < a.c0;
  const int ch0 = min(ch, a.c0 - 4);
  f32x4 v;
  if (SRC == PSM_SRC_SAME) {
    v = read4<KSM>(in0 + ((int64_t)yc * a.P0 + xc) * a.c0 + ch0, a.ks0, a.slab0, a.pbias0, ch0);
  } else if (SRC == PSM_SRC_UPSAMPLE) {
    v = read4<KSM>(in0 + ((int64_t)(yc >> 1) * a.P0 + (xc >> 1)) * a.c0 + ch0, a.ks0, a.slab0, a.pbias0, ch0);
  } else {
    const float* p = in0 + ((int64_t)(2 * yc) * a.P0 + 2 * xc) * a.c0 + ch0;
    const f32x4 q0 = read4<KSM>(p, a.ks0, a.slab0, a.pbias0, ch0);
    const f32x4 q1 = read4<KSM>(p + a.c0, a.ks0, a.slab0, a.pbias0, ch0);
    const f32x4 q2 = read4<KSM>(p + (int64_t)a.P0 * a.c0, a.ks0, a.slab0, a.pbias0, ch0);
    const f32x4 q3 = read4<KSM>(p + (int64_t)a.P0 * a.c0 + a.c0, a.ks0, a.slab0, a.pbias0, ch0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaxf(q0[j], q1[j]), fmaxf(q2[j], q3[j]));
  }
  bool ok = inside && first;
  if (SRC == PSM_SRC_UPSAMPLE) {                      // the only source with a skip input concatenated behind it
    const int c = min(max(ch - a.c0, 0), a.c1 - 4);
    const f32x4 s = read4<KSM>(in1 + ((int64_t)yc * a.P1 + xc) * a.c1 + c, a.ks1, a.slab1, a.pbias1, c);
    const bool second = !first && (ch - a.c0) < a.c1;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = first ? v[j] : s[j];
    ok = inside && (first || second);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0.f;
  return v;
}

// stem (c_in not a multiple of 4, e.g. the 3-channel grid image): scalar loads, same-resolution source only
__device__ __forceinline__ f32x4 fetch4_stem(const PsmConvArgs& a, const float* in0, int y, int x, int ch) {
  const bool inside = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
  const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
  const float* p = in0 + ((int64_t)yc * a.P0 + xc) * a.c0;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = p[min(ch + j, a.c0 - 1)];
    v[j] = (inside && ch + j < a.c0) ? t : 0.f;
  }
  return v;
}

// Loop-invariant part of one 4-channel fetch (pixel position -> element offsets, padding predicate), computed once
// per workgroup: inside the chunk loop a fetch is then one add and one (slab-summed) 16-byte load.  The address
// arithmetic would otherwise run again for every chunk, in front of the MFMAs (~600 VALU instructions per chunk).
struct PsmFetchPos {
  int off0;      // element offset of channel 0 of the source pixel in in0 (for max-pool: its top-left pixel)
  int off1;      // same in the skip input
  bool ok;       // inside the image (else the zero padding)
};

template <int SRC>
__device__ __forceinline__ PsmFetchPos prepare_fetch(const PsmConvArgs& a, int y, int x) {
  PsmFetchPos f;
  f.ok = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
  const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
  if (SRC == PSM_SRC_UPSAMPLE) f.off0 = ((yc >> 1) * a.P0 + (xc >> 1)) * a.c0;
  else if (SRC == PSM_SRC_MAXPOOL) f.off0 = ((2 * yc) * a.P0 + 2 * xc) * a.c0;
  else f.off0 = (yc * a.P0 + xc) * a.c0;
  f.off1 = (yc * a.P1 + xc) * a.c1;
  return f;
}

// channels [ch, ch+4) of a chunk that lies entirely in in0 (from0, uniform) or entirely in the skip input.  The
// source is chosen with selects on pointer / slab parameters -- no branch around the loads.
template <int SRC, int KSM>
__device__ __forceinline__ f32x4 fetch4_prepared(const PsmConvArgs& a, const float* in0, const float* in1, const PsmFetchPos& f,
                                                 int ch, bool from0) {
  f32x4 v;
  if (SRC == PSM_SRC_MAXPOOL) {
    const float* p = in0 + f.off0 + ch;
    const f32x4 q0 = read4<KSM>(p, a.ks0, a.slab0, a.pbias0, ch);
    const f32x4 q1 = read4<KSM>(p + a.c0, a.ks0, a.slab0, a.pbias0, ch);
    const f32x4 q2 = read4<KSM>(p + a.P0 * a.c0, a.ks0, a.slab0, a.pbias0, ch);
    const f32x4 q3 = read4<KSM>(p + a.P0 * a.c0 + a.c0, a.ks0, a.slab0, a.pbias0, ch);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaxf(q0[j], q1[j]), fmaxf(q2[j], q3[j]));
  } else if (SRC == PSM_SRC_UPSAMPLE) {
    const float* p = from0 ? in0 + f.off0 + ch : in1 + f.off1 + (ch - a.c0);
    const int ks = from0 ? a.ks0 : a.ks1;
    const int64_t slab = from0 ? a.slab0 : a.slab1;
    const float* pb = from0 ? a.pbias0 : a.pbias1;
    v = read4<KSM>(p, ks, slab, pb, from0 ? ch : ch - a.c0);
  } else {
    v = read4<KSM>(in0 + f.off0 + ch, a.ks0, a.slab0, a.pbias0, ch);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = f.ok ? v[j] : 0.f;
  return v;
}

// raw loads of one 4-channel group (all slabs), and their combination: split so that the loads of chunk g+1 can stay
// in flight during the MFMAs of chunk g -- a combine (or even a zeroing select) placed right behind the loads makes the
// compiler wait for them BEFORE the MFMAs, i.e. one exposed memory round trip per chunk
template <int KSM>
__device__ __forceinline__ void issue4(const float* p, int ks, int64_t slab, f32x4* r) {
  r[0] = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int s = 1; s < KSM; ++s) r[s] = *reinterpret_cast<const f32x4*>(p + (int64_t)min(s, ks - 1) * slab);
}
template <int KSM>
__device__ __forceinline__ f32x4 combine4(const f32x4* r, int ks, f32x4 b) {      // same order as read4
  f32x4 v = r[0];
  if (KSM > 1) {
#pragma unroll
    for (int s = 1; s < KSM; ++s) v += r[s] * (s < ks ? 1.f : 0.f);
    const bool fin = ks > 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fin ? fmaxf(v[j] + b[j], 0.f) : v[j];
  }
  return v;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// element-wise maximum of two groups of 8 packed bf16 (16 bytes each).  The 2 x 2 max-pool only ever reads outputs of ReLU'd convolutions
// (every 3x3 layer has one; the linear head feeds no pool): NON-NEGATIVE values, whose bf16 bit patterns order like signed 16-bit integers
// (-0 below +0, a positive NaN above everything, so it propagates like np.maximum) -- four v_pk_max_i16 instead of the 36 shift / mask /
// float-max / pack instructions of the float form, three times per staged piece (round 6: the pool staging of the encoder launches is
// vector-instruction bound).
__device__ __forceinline__ f32x4 bf16x8_max(f32x4 a, f32x4 b) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
}
// activation store of one value per lane (lane & 15 = channel): float32, or bf16 with the channel pair (c, c+1) packed
// into one 4-byte store by the even lane (the odd neighbour's value arrives through a DPP quad permute)
__device__ __forceinline__ void store_act(float* out, int64_t elem, float v, bool ok, bool out_bf, int co) {
  if (out_bf) {
    const float vn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)vn) << 16);
    if (ok && !(co & 1)) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(out) + elem) = pk;
  } else if (ok) {
    PSM_ACT_STORE(&out[elem], v);
  }
}

// TH: tile rows; WM: rows per wave; NCT: channel tiles per workgroup; WN: channel tiles per wave; BF: bf16 operands.
// Software pipeline over the channel chunks, both operands double-buffered in LDS: the input tile (through the
// source transform) and the weights (9 x NCT KiB, already in MFMA operand order) of chunk g+1 are requested
// into registers before the MFMAs of chunk g, combined (slab sums, max-pool, padding selects) and written to the
// other LDS buffers after them; one LDS-only barrier per chunk.  Inside a chunk the LDS operands of tap t+1 are
// read before the MFMAs of tap t are issued.  Cooperative staging keeps the register count low, so that two
// workgroups share a CU and hide each other's prologue.
//   f32 (BF = false): chunks of 16 channels, v_mfma_f32_16x16x4_f32, exact f32 products.
//   bf16 (BF = true): activations (after the source transform) and weights rounded to bf16 (RNE), chunks of 32
//     channels, ONE v_mfma_f32_16x16x32_bf16 per (tap, row, channel tile): lane l holds A[pixel l&15][k = 8*(l>>4)+j]
//     and B[k][channel l&15], j < 8.  Without ABF activations are float32 in HBM (skips, slabs and the oracle's rounding
//     points are unchanged).  Both forms use a 64-byte LDS pixel with swizzled 16-byte slots (lds_slot).
//   NB: LDS buffers per operand -- 1 when a workgroup has a single chunk (nothing to pipeline: half the LDS, twice
//     the workgroups per CU to cover each other's load latency), else 2.
//   ABF (bf16 mode, finished inputs only): the inputs are bf16 in HBM; a fetch is 8 channels = 16 bytes that go to LDS as
//     they are (max-pool: element-wise max of the four source pixels first) -- half the fetches, no conversion.
//   X6 (float32 mode, round 3): float32 arithmetic on the bf16 matrix pipe.  Activations (on their way into LDS) and
//     weights (on the host) are split EXACTLY into three bf16 planes x = hi + mid + lo (8 + 8 + 8 significant bits); a
//     product runs as the six MFMA terms hh, hm, mh, hl, lh, mm (what is dropped is below 2^-24 of the product), small
//     terms first, float32 accumulation: 6 x 16 cycles per 32 channels against 8 x 32 for v_mfma_f32_16x16x4_f32.  Chunks
//     of 32 channels, three LDS planes per operand; the weights (27 KiB x NCT per chunk) are single-buffered.
//   KW = 2 (bf16 activations, two or more chunks; PsmConvArgs::kw): IN-WORKGROUP K SPLIT.  A workgroup of eight waves; waves 0-3 walk the first half of
//     the workgroup's channel chunks, waves 4-7 the second half, each half with its own pair of LDS buffers per operand (the halves share
//     the chunk barriers: same chunk count, a make-up barrier where it differs by one); the accumulators of the second half meet the
//     first half's through LDS behind the last chunk, waves 0-3 run the epilogue.  Half as long a chain of dependent chunks per wave and
//     two waves per SIMD on a chip the launch cannot fill with workgroups -- what a split over workgroups (ksplit) buys, without
//     float32 partial-sum slabs for the consumer to add up.
template <int TH, int WM, int NCT, int WN, int SRC, int KSM, bool BF, int NB, bool ABF = false, bool X6 = false, int KW = 1>     // SRC: PSM_SRC_*, -1 = unaligned stem, 3 = upsample + skip with the seam inside a chunk
__global__ __launch_bounds__(256 * KW) void psm_conv3x3_kernel(PsmConvArgs a, int co_groups) {
  static_assert(KW == 1 || (KW == 2 && ABF && NB == 2 && !X6), "in-workgroup K split: bf16 activations, double-buffered chunks");
  static_assert(!ABF || (BF && KSM == 1 && SRC >= 0 && SRC != 3), "bf16 activations: finished same / upsample / max-pool sources only");
  static_assert(!X6 || (BF && !ABF && NB == 2 && SRC >= 0), "x6: float32 inputs, bf16 MFMA, double-buffered input planes");
  constexpr int PL = X6 ? 3 : 1;                                    // operand planes
  constexpr int NBW = X6 ? 1 : NB;                                  // weight buffers
  constexpr int CB = BF ? 32 : 16;                                  // input channels per chunk
  constexpr int FG = ABF ? 8 : 4;                                   // channels per fetch (16 bytes: 4 float32 or 8 bf16)
  constexpr int G4 = CB / FG;                                       // fetch groups per pixel
  constexpr int NPIX = (TH + 2) * (TW + 2);
  constexpr int NF = (NPIX * G4 + 255) / 256;                       // 4-channel input fetches per thread and chunk
  constexpr int WQ = PL * 9 * NCT * 64;                             // 16-byte pieces per weight chunk (x6: [plane][tap][ct][lane])
  constexpr int NWF = (WQ + 255) / 256;
  // both LDS buffers are rounded up to whole fetch rounds: every thread stores every round (the surplus lands in the
  // pad), so no store sits behind a branch -- a skipped store leaves its load "pending" for the compiler, which then
  // drains vmcnt before the register's next load, in the middle of the MFMA stream
  constexpr int TILE = (NF * 256 / G4) * LDC;                       // floats per input-tile buffer (64 B per pixel)
  constexpr int WQP = NWF * 256;                                    // weight-buffer stride (16-byte pieces)
  constexpr int SRCP = SRC == 3 ? PSM_SRC_UPSAMPLE : (SRC < 0 ? 0 : SRC);
  constexpr int NQ = SRCP == PSM_SRC_MAXPOOL ? 4 : 1;               // source pixels per fetch
  constexpr bool DEFER = ABF || (SRC >= 0 && SRC != 3 && NF * NQ * KSM <= 24);   // raw loads held across the MFMAs (<= 96 VGPRs)
  constexpr int NRAW = DEFER ? NQ * KSM : 1;
  __shared__ __attribute__((aligned(16))) float in_tile[KW * NB * PL * TILE];
  __shared__ __attribute__((aligned(16))) f32x4 w_tile[KW * NBW * WQP];
  static_assert(KW == 1 || (size_t)WM * WN * 256 <= (size_t)KW * NBW * WQP, "the second half's accumulators fit the weight buffers");
  const int tid = KW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255), lane = tid & 63;     // position within the half
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = KW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int hin = half * (NB * PL * TILE), hw = half * (NBW * WQP);                             // the half's LDS areas (floats / 16-byte pieces)
  const int zz = blockIdx.z / a.ksplit, split = blockIdx.z - zz * a.ksplit;
  const int cs = zz / co_groups, cog = zz - cs * co_groups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* in0 = a.in0 + (int64_t)cs * a.in0_case;
  const float* in1 = a.in1 ? a.in1 + (int64_t)cs * a.in1_case : nullptr;
  const int row_w = (TH == 4 * WM) ? wave * WM : 0;           // pixel-major: waves stacked along the rows
  const int ct_w = (TH == 4 * WM) ? 0 : wave * WN;            // channel-major: waves along the channel tiles
  const int px = lane & 15, kq = lane >> 4;
  f32x4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the epilogue's bias, requested HERE with the first chunk's operands: loaded in the epilogue it was an exposed round trip
  // at the end of every workgroup (stamps: last barrier -> end 1.5 us of a 5 us workgroup on the 64^2 layers)
  // (the bias array is zero-padded to whole channel tiles and four tiles beyond: upload_conv)
  f32x4 bias_r[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) bias_r[n] = *reinterpret_cast<const f32x4*>(a.bias + (cog * NCT + ct_w + n) * 16 + 4 * kq);
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpack) + (int64_t)cog * a.n_chunks * WQ;
  const int cps = (a.n_chunks + a.ksplit - 1) / a.ksplit;          // chunks per split
  const int g_beg0 = split * cps, g_end0 = min(a.n_chunks, (split + 1) * cps);
  const int hc = KW == 1 ? g_end0 - g_beg0 : (g_end0 - g_beg0 + 1) / 2;           // chunks of a half (the first half's count)
  const int g_beg = g_beg0 + half * hc, g_end = min(g_end0, g_beg + hc);

  f32x4 xr[NF][NRAW], xb[NF], wr[NWF];
  // (uniform chunk base + this thread's piece, which is the same for every chunk: a scalar add and no vector address arithmetic per request)
  unsigned wpiece[NWF];
#pragma unroll
  for (int u = 0; u < NWF; ++u) { wpiece[u] = (unsigned)min(tid + 256 * u, WQ - 1); asm volatile("" : "+v"(wpiece[u])); }
  auto load_w_one = [&](int g, int u) { wr[u] = (wsrc + (int64_t)g * WQ)[wpiece[u]]; };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NWF; ++u) w_tile[hw + buf * WQP + tid + 256 * u] = wr[u];
  };
  // loop-invariant fetch positions; chunks normally lie on one side of the concatenation seam (channel counts
  // are multiples of the chunk), else the general per-lane path is taken
  PsmFetchPos fp[NF];
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const int pos = (tid + 256 * u) / G4;
    const int r = min(pos / (TW + 2), TH + 1), c = pos - (pos / (TW + 2)) * (TW + 2);
    fp[u] = prepare_fetch<SRCP>(a, y0 - 1 + r, x0 - 1 + c);
  }
  // request chunk g: straight-line loads only (clamped addresses); what needs the data comes in finish_x
  auto issue_x_one = [&](int g, int u) {
    const bool from0 = g * CB < a.c0;                 // uniform
    const int lim = from0 ? a.c0 : a.c0 + a.c1;
    {
      const int q = tid + 256 * u;
      const int pos = q / G4, c4 = q & (G4 - 1);
      const int r = pos / (TW + 2), c = pos - r * (TW + 2);
      const int rr = min(r, TH + 1);
      const int ch = g * CB + FG * c4;
      const int chc = min(ch, lim - FG);    // channels beyond the real inputs (zero-padded tail): clamped, zeroed later
      if constexpr (ABF) {                  // 8 bf16 channels per 16-byte load, element offsets in halves
        const unsigned short* h0 = reinterpret_cast<const unsigned short*>(a.in0) + (int64_t)cs * a.in0_case;
        const unsigned short* h1 = reinterpret_cast<const unsigned short*>(a.in1) + (int64_t)cs * a.in1_case;
        // a wave-uniform base pointer + an unsigned 32-bit element offset per lane (offsets into a case's tensor are non-negative and
        // below 2^31: act_layout): the scalar-base form of global_load, one or two vector instructions per request instead of a 64-bit chain
        if constexpr (SRCP == PSM_SRC_MAXPOOL) {
          const unsigned e = (unsigned)(fp[u].off0 + chc);
          xr[u][0] = *reinterpret_cast<const f32x4*>(h0 + e);
          xr[u][1] = *reinterpret_cast<const f32x4*>(h0 + (e + (unsigned)a.c0));
          xr[u][2] = *reinterpret_cast<const f32x4*>(h0 + (e + (unsigned)(a.P0 * a.c0)));
          xr[u][3] = *reinterpret_cast<const f32x4*>(h0 + (e + (unsigned)(a.P0 * a.c0 + a.c0)));
        } else if constexpr (SRCP == PSM_SRC_UPSAMPLE) {
          const unsigned short* base = from0 ? h0 : h1;                                   // uniform
          const unsigned e = (unsigned)(from0 ? fp[u].off0 + chc : fp[u].off1 + (chc - a.c0));
          xr[u][0] = *reinterpret_cast<const f32x4*>(base + e);
        } else {
          xr[u][0] = *reinterpret_cast<const f32x4*>(h0 + (unsigned)(fp[u].off0 + chc));
        }
      }
      else if constexpr (SRC < 0) xr[u][0] = fetch4_stem(a, in0, y0 - 1 + rr, x0 - 1 + c, ch);
      else if constexpr (SRC == 3) xr[u][0] = fetch4<PSM_SRC_UPSAMPLE, KSM>(a, in0, in1, y0 - 1 + rr, x0 - 1 + c, ch);
      else if constexpr (!DEFER) {
        const f32x4 t = fetch4_prepared<SRCP, KSM>(a, in0, in1, fp[u], chc, from0);
        const bool live = ch < lim;
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[u][0][j] = live ? t[j] : 0.f;
      } else if constexpr (SRCP == PSM_SRC_MAXPOOL) {
        const float* p = in0 + fp[u].off0 + chc;
        issue4<KSM>(p, a.ks0, a.slab0, &xr[u][0]);
        issue4<KSM>(p + a.c0, a.ks0, a.slab0, &xr[u][KSM]);
        issue4<KSM>(p + a.P0 * a.c0, a.ks0, a.slab0, &xr[u][2 * KSM]);
        issue4<KSM>(p + a.P0 * a.c0 + a.c0, a.ks0, a.slab0, &xr[u][3 * KSM]);
        if (KSM > 1) xb[u] = *reinterpret_cast<const f32x4*>(a.pbias0 + chc);
      } else if constexpr (SRCP == PSM_SRC_UPSAMPLE) {
        const float* p = from0 ? in0 + fp[u].off0 + chc : in1 + fp[u].off1 + (chc - a.c0);
        issue4<KSM>(p, from0 ? a.ks0 : a.ks1, from0 ? a.slab0 : a.slab1, &xr[u][0]);
        if (KSM > 1) xb[u] = *reinterpret_cast<const f32x4*>(from0 ? a.pbias0 + chc : a.pbias1 + (chc - a.c0));
      } else {
        issue4<KSM>(in0 + fp[u].off0 + chc, a.ks0, a.slab0, &xr[u][0]);
        if (KSM > 1) xb[u] = *reinterpret_cast<const f32x4*>(a.pbias0 + chc);
      }
    }
  };
  // the NI = NF + NWF requests of a chunk are spread over the nine taps of the previous chunk's MFMA phase: a wave-wide
  // 16-byte load occupies the CU's address path for 16 cycles (64 B/clk), so a chunk's 30-130 loads issued in one go
  // hold the waves -- and the MFMAs behind them in program order -- for 0.3-1 us
  constexpr int NI = NF + NWF;
  auto issue_item = [&](int g, int i) {
    if (i < NF) issue_x_one(g, i); else load_w_one(g, i - NF);
  };
  // combine what issue_x_one requested (slab sums + producer bias / ReLU, 2x2 max, zero padding) and write the LDS tile
  auto finish_x = [&](int g, int buf) {
    const bool from0 = g * CB < a.c0;
    const int lim = from0 ? a.c0 : a.c0 + a.c1;
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int q = tid + 256 * u;
      const int pos = q / G4, c4 = q & (G4 - 1);
      f32x4 v = xr[u][0];
      if constexpr (ABF) {
        if constexpr (NQ == 4) v = bf16x8_max(bf16x8_max(xr[u][0], xr[u][1]), bf16x8_max(xr[u][2], xr[u][3]));
        const bool ok = fp[u].ok && (g * CB + FG * c4 < lim);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0.f;
        *reinterpret_cast<f32x4*>(&in_tile[hin + buf * TILE + lds_slot(pos, c4)]) = v;      // 8 bf16 = one 16-byte slot of the pixel
        continue;
      }
      if constexpr (DEFER) {
        const int ks = (SRCP == PSM_SRC_UPSAMPLE && !from0) ? a.ks1 : a.ks0;
        v = combine4<KSM>(&xr[u][0], ks, xb[u]);
        if constexpr (NQ == 4) {
          const f32x4 q1 = combine4<KSM>(&xr[u][KSM], ks, xb[u]), q2 = combine4<KSM>(&xr[u][2 * KSM], ks, xb[u]),
                      q3 = combine4<KSM>(&xr[u][3 * KSM], ks, xb[u]);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaxf(v[j], q1[j]), fmaxf(q2[j], q3[j]));
        }
        const bool ok = fp[u].ok && (g * CB + FG * c4 < lim);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0.f;
      }
      if constexpr (X6) {
        const bf16x4 h = __builtin_convertvector(v, bf16x4);                  // round to nearest even
        const f32x4 r1 = v - __builtin_convertvector(h, f32x4);               // exact
        const bf16x4 md = __builtin_convertvector(r1, bf16x4);
        const f32x4 r2 = r1 - __builtin_convertvector(md, f32x4);             // exact, <= 8 significant bits
        const bf16x4 lo = __builtin_convertvector(r2, bf16x4);
        __bf16* dst = reinterpret_cast<__bf16*>(&in_tile[buf * PL * TILE + lds_slot(pos, c4 >> 1)]) + 4 * (c4 & 1);
        *reinterpret_cast<bf16x4*>(dst) = h;
        *reinterpret_cast<bf16x4*>(dst + 2 * TILE) = md;                      // plane stride TILE floats = 2 * TILE bf16
        *reinterpret_cast<bf16x4*>(dst + 4 * TILE) = lo;
      } else if constexpr (BF) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(&in_tile[buf * TILE + lds_slot(pos, c4 >> 1)]) + 4 * (c4 & 1)) = h;
      } else {
        *reinterpret_cast<f32x4*>(&in_tile[buf * TILE + lds_slot(pos, c4)]) = v;
      }
    }
  };

  USTAMP(0);
  if (g_beg < g_end) {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_item(g_beg, i);
  }
  __builtin_amdgcn_sched_barrier(0);             // the first chunk's requests are in flight: the slot arithmetic below runs under their latency
  // LDS slots of the activation operand reads: (WM + 2) tile rows x 3 tap columns, the same for every chunk and buffer (round 6).  hipcc
  // recomputed the swizzled slot -- five vector instructions -- in front of each of a chunk's 9 WM reads, inside a matrix phase whose 36
  // MFMAs the wave's own instruction stream, not the matrix pipe, paces (stamps: 0.60 us per chunk for 0.27 us of MFMA issue); the
  // empty asm keeps the values in registers instead of rematerialising them.
  int a_slot[WM + 2][3];
#pragma unroll
  for (int r = 0; r < WM + 2; ++r)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      a_slot[r][kx] = hin + lds_slot((row_w + r) * (TW + 2) + px + kx, kq);
      asm volatile("" : "+v"(a_slot[r][kx]));
    }
  if (g_beg < g_end) {
    finish_x(g_beg, 0);
    store_w(0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  USTAMP(1);
  // (the LDS buffer of a chunk is a COMPILE-TIME constant of its code copy -- the chunk loop is unrolled by two -- so that every LDS read and
  // write of the matrix phase is a register slot + an immediate offset: with a run-time buffer index each read cost a vector add; round 6)
  // one chunk; MORE (compile time): the next chunk's requests are issued among the taps and stored after them.  The
  // last chunk is a second copy of the body rather than a run-time `more` flag: requests behind a condition the
  // compiler cannot correlate from tap to tap get a full vmcnt(0) drain in front of each of them
  auto run_chunk = [&](int g, auto more_tag, auto buf_tag) {
    constexpr bool more = decltype(more_tag)::value;
    constexpr int buf = decltype(buf_tag)::value;
    USTAMP(2 + 4 * (g - g_beg));
    const float* tile = &in_tile[buf * PL * TILE];
    const f32x4* wt = &w_tile[hw + (NBW == 2 ? buf : 0) * WQP + ct_w * 64 + lane];
    // operand prefetch depth in taps (-DPSM_CONV_PD=n, bf16 form).  Measured with 3 instead of 1: nothing (8 cases bf16 158.4 vs
    // 157.5 us, dec3a 14.6 vs 14.5): a bf16 chunk's matrix phase (0.57 us for 36 MFMAs = 0.24 us of issue) is bound by the
    // THROUGHPUT of its 36 ds_read_b128 per wave (four waves: 0.48 us), not by their latency
    if constexpr (BF && !X6 && PSM_CONV_KYREUSE) {
      // bf16 form: ACTIVATION ROWS KEPT ACROSS ky.  The matrix phase of a bf16 chunk was bound by the throughput of its LDS
      // operand reads (one ds_read_b128 per 16-cycle MFMA: 36 per wave for 36 MFMAs).  For a tap column kx the WM rows of the
      // three ky taps are WM + 2 distinct tile rows: they are read once and used by all three -- (WM + 2) + 3 WN reads per
      // 3 WM WN MFMAs (WM = WN = 2: 10 per 12; WM = 8, WN = 1: 13 per 24).  The rows and weights of column kx + 1 are read
      // while the MFMAs of column kx issue.  Accumulation order: (kx, ky) instead of (ky, kx).
      f32x4 avr[2][WM + 2], bvr[2][3][WN];
      auto lds_read_kx = [&](int kx, int s2) {
#pragma unroll
        for (int r = 0; r < WM + 2; ++r)
          avr[s2][r] = *reinterpret_cast<const f32x4*>(&tile[a_slot[r][kx]]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int n = 0; n < WN; ++n) bvr[s2][ky][n] = wt[((ky * 3 + kx) * NCT + n) * 64];
      };
      lds_read_kx(0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int s2 = kx & 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int step = kx * 3 + ky;
          if (ky == 0 && kx + 1 < 3) lds_read_kx(kx + 1, s2 ^ 1);
#if !defined(PSM_EXP) || PSM_EXP == 3
          if constexpr (more) {                            // this step's share of the next chunk's requests
#pragma unroll
            for (int i = 0; i < NI; ++i)
              if ((i * 9) / NI == step) issue_item(g + 1, i);
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
#if !defined(PSM_EXP) || PSM_EXP < 3
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bvr[s2][ky][n]), __builtin_bit_cast(bf16x8, avr[s2][m + ky]),
                                                                  acc[m][n], 0, 0, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    constexpr int PD = (BF && !X6) ? PSM_CONV_PD : 1, NS = PD + 1;
    f32x4 av[NS][WM][PL], bv[NS][WN][PL];
    auto lds_read = [&](int tap, int s) {
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int p = 0; p < PL; ++p)
          av[s][m][p] = *reinterpret_cast<const f32x4*>(&tile[p * TILE + a_slot[m + ky][kx]]);
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int p = 0; p < PL; ++p) bv[s][n][p] = wt[((p * 9 + tap) * NCT + n) * 64];
    };
#pragma unroll
    for (int t0 = 0; t0 < PD; ++t0) lds_read(t0, t0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = tap % NS;
      if (tap + PD < 9) lds_read(tap + PD, (tap + PD) % NS);      // operands of the tap PD ahead are on their way during this tap's MFMAs
#if !defined(PSM_EXP) || PSM_EXP == 3
      if constexpr (more) {                            // this tap's share of the next chunk's requests
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if ((i * 9) / NI == tap) issue_item(g + 1, i);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !defined(PSM_EXP) || PSM_EXP < 3
      if constexpr (X6) {
        // (activation plane, weight plane) pairs, small terms first: mm, lh, hl, mh, hm, hh
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv[s][n][PB[t6]]),
                                                                  __builtin_bit_cast(bf16x8, av[s][m][PA[t6]]), acc[m][n], 0, 0, 0);
      } else if constexpr (BF) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv[s][n][0]), __builtin_bit_cast(bf16x8, av[s][m][0]),
                                                                acc[m][n], 0, 0, 0);
      } else {
        // consecutive MFMAs go to different accumulators (dependent-accumulator latency 40 > issue 32 cycles)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) acc[m][n] = MFMA16(bv[s][n][0][j], av[s][m][0][j], acc[m][n]);
      }
#endif
      // nothing may move across: above all not the combines / LDS stores below, which wait for the loads issued above
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    USTAMP(3 + 4 * (g - g_beg));
#if !defined(PSM_EXP) || PSM_EXP == 3
    if constexpr (more) {
      finish_x(g + 1, buf ^ 1);
      if constexpr (NBW == 2) store_w(buf ^ 1);
      else {                           // single weight buffer: every wave has read chunk g's fragments before they are replaced
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        store_w(0);
      }
    }
#endif
    USTAMP(4 + 4 * (g - g_beg));
#if !defined(PSM_EXP) || PSM_EXP != 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    USTAMP(5 + 4 * (g - g_beg));
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if constexpr (NB == 2) {
    int g = g_beg;
    for (; g + 2 < g_end; g += 2) { run_chunk(g, std::true_type{}, B0{}); run_chunk(g + 1, std::true_type{}, B1{}); }
    if (g + 1 < g_end) { run_chunk(g, std::true_type{}, B0{}); run_chunk(g + 1, std::false_type{}, B1{}); }
    else if (g < g_end) run_chunk(g, std::false_type{}, B0{});
  } else {
    if (g_beg < g_end) run_chunk(g_end - 1, std::false_type{}, B0{});
  }
  if constexpr (KW == 2) {
    // the halves share the workgroup barrier: the second half makes up for a chunk it does not have (odd chunk counts)
    for (int i = g_end - g_beg; i < hc; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // K halves meet: behind the last chunk's barrier nobody reads the weight buffers any more
    f32x4* red = w_tile;
    if (half == 1) {
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) red[(m * WN + n) * 256 + tid] = acc[m][n];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (half == 1) return;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) acc[m][n] += red[(m * WN + n) * 256 + tid];
  }
  // ---- epilogue: bias + ReLU (split-K: the raw partial sum into this split's slab), NHWC store
  float* out = a.out_bf ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(a.out) + (int64_t)cs * a.out_case)
                        : a.out + (int64_t)cs * a.out_case + (int64_t)split * a.out_slab;
  // Stores are write-through (sc1, PSM_WT_STORES): see store_act8 in psm_unet_pair.hip -- the dirty lines of an activation would be
  // written back in one burst at the end of the kernel anyway (the consumers run on other XCDs).
  // The MFMAs take the WEIGHTS as their first operand: D[channel][pixel], i.e. lane l holds the four consecutive channels
  // 4 * (l >> 4) .. + 3 of pixel l & 15 -- one 8-byte (bf16) or 16-byte (float32) store per (row, channel tile) and one address per
  // lane.  (Round 4's layout, a lane = four pixels of ONE channel, needed 16 predicated 4-byte stores with a 64-bit address chain
  // each: stamps showed 1.5 us between the last barrier and the end of a 5 us workgroup.)
  const bool fin = a.ksplit == 1;
  const int x = x0 + px;
  const bool xok = x < a.W;
  const int64_t pix0 = ((int64_t)(y0 + row_w) * a.PO + x) * a.cout;
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int co4 = (cog * NCT + ct_w + n) * 16 + 4 * kq;
    const f32x4 b = fin ? bias_r[n] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      f32x4 v = acc[m][n] + b;
      if (fin && a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      acc[m][n] = v;
      const bool ok = xok && y0 + row_w + m < a.H;
      const int64_t e = pix0 + (int64_t)m * a.PO * a.cout + co4;
      if ((a.cout & 3) == 0) {
        if (ok && co4 < a.cout) {
          if (a.out_bf) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
            u32x2 pk;
            pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
#if PSM_WT_STORES
            asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(reinterpret_cast<unsigned short*>(out) + e), "v"(pk) : "memory");
          } else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(out + e), "v"(v) : "memory");
#else
            *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + e) = pk;
          } else *reinterpret_cast<f32x4*>(out + e) = v;
#endif
        }
      } else {                                              // channel counts that are not multiples of four: element-wise
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ok && co4 + r < a.cout) {
            if (a.out_bf) reinterpret_cast<unsigned short*>(out)[e + r] = __builtin_bit_cast(unsigned short, (__bf16)v[r]);
            else out[e + r] = v[r];
          }
      }
    }
  }
  if constexpr (NCT == 1 && WN == 1) {
    if (a.head_w) {                                                         // uniform branch, every lane active
      f32x4 hv[WM];
#pragma unroll
      for (int m = 0; m < WM; ++m) hv[m] = acc[m][0];
      head_epilogue<WM>(a, cs, y0 + row_w, x0, lane, hv);
    }
  }
  USTAMP(63);
}

// ---------------------------------------------------------------------------------------------------
// stem: the first layer reads the raw grid image (c_in = 3: not a multiple of 4, and only 27 contraction
// terms).  Padding its three channels to a chunk of 16 would run 144-term MFMAs for 27 useful terms; here
// K = 9 * c_in is flattened (k = tap * c_in + channel) and padded to KG groups of 16.  The (8+2) x (16+2) x c_in
// image tile goes to LDS with row-contiguous loads (every load issued up front, clamped + selected); the A
// operand of pixel p, term k is then the LDS word at tile[p + (ky, kx)][channel] -- four ds_read_b32 per
// k-group, addresses fixed per lane -- and the padded terms read a zero word.  Weights go from global memory to
// registers in MFMA order.  bf16 mode: operands rounded to bf16 first (products exact in the f32 MFMA).
// ---------------------------------------------------------------------------------------------------
// C0 > 0: the channel count as a compile-time constant (3 for the reference's grid image): the per-element index
// arithmetic (e / c0, k / c0, ...) is then multiplications instead of ~1000 VALU instructions of integer division per
// thread, which -- with 8 waves per SIMD -- cost more than the layer's memory traffic; C0 == 0: run-time count.
template <int KG, int C0>
__global__ __launch_bounds__(256) void psm_conv_stem_kernel(PsmConvArgs a) {
  constexpr int TH = 8, NPIX = (TH + 2) * (TW + 2), CMAX = 7, NE = (NPIX * CMAX + 255) / 256, ZERO = NE * 256;
  __shared__ float tile[ZERO + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cs = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* in0 = a.in0 + (int64_t)cs * a.in0_case;
  const int c0 = C0 > 0 ? C0 : a.c0, K = 9 * c0, nval = NPIX * c0;
  float ev[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = min(tid + 256 * u, nval - 1), pos = e / c0, ci = e - pos * c0;
    const int r = pos / (TW + 2), c = pos - r * (TW + 2);
    const int y = y0 - 1 + r, x = x0 - 1 + c;
    const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
    const float t = in0[((int64_t)min(max(y, 0), a.H - 1) * a.P0 + min(max(x, 0), a.W - 1)) * c0 + ci];
    ev[u] = ok ? t : 0.f;
  }
  f32x4 bw[KG];
  const float4* wsrc = a.wpack + lane;
#pragma unroll
  for (int g = 0; g < KG; ++g) { const float4 t = wsrc[g * 64]; bw[g] = (f32x4){t.x, t.y, t.z, t.w}; }
  const float bias = a.bias[lane & 15];
  const int px = lane & 15, kq = lane >> 4;
  int off[KG][4];                               // LDS word of term k relative to the pixel, or the zero word
#pragma unroll
  for (int g = 0; g < KG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * g + 4 * kq + j, kc = min(k, K - 1), tap = kc / c0, ci = kc - tap * c0;
      const int ky = tap / 3, kx = tap - 3 * ky;
      off[g][j] = k < K ? (ky * (TW + 2) + kx) * c0 + ci : -1;
    }
#pragma unroll
  for (int u = 0; u < NE; ++u) tile[tid + 256 * u] = a.bf16 ? (float)(__bf16)ev[u] : ev[u];     // surplus rounds land in the pad
  if (tid == 0) tile[ZERO] = 0.f;
  __syncthreads();
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    float av[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int base = ((2 * wave + m) * (TW + 2) + px) * c0;
#pragma unroll
      for (int j = 0; j < 4; ++j) av[m][j] = tile[off[g][j] < 0 ? ZERO : base + off[g][j]];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = MFMA16(av[m][j], bw[g][j], acc[m]);
  }
  float* out = a.out_bf ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(a.out) + (int64_t)cs * a.out_case) : a.out + (int64_t)cs * a.out_case;
  const int co = lane & 15;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int y = y0 + 2 * wave + m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = x0 + 4 * kq + r;
      float v = acc[m][r] + bias;
      if (a.relu) v = fmaxf(v, 0.f);
      store_act(out, ((int64_t)y * a.PO + x) * a.cout + co, v, y < a.H && x < a.W && co < a.cout, a.out_bf != 0, co);
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

template <int TH, int WM, int NCT, int WN>
static void launch_variant(const PsmConvArgs& a, dim3 grid, int groups, hipStream_t st) {
  const bool stem = (a.c0 % 4 != 0) || (a.c1 % 4 != 0);
  const bool slabs = a.ks0 > 1 || a.ks1 > 1;
  const bool one = (a.n_chunks + a.ksplit - 1) / a.ksplit <= 1;      // a single chunk per workgroup: single LDS buffers
#define GO(S, K)                                                                                                         \
  do {                                                                                                                   \
    if constexpr (K == 1 && S >= 0 && S != 3) {                                                                          \
      if (a.bf16 && a.in_bf) {                                                                                           \
        if constexpr (TH == 8 && NCT <= 2) {       /* in-workgroup K split (PsmConvArgs::kw): eight waves, the 8-row tiles only (LDS) */ \
          if (a.kw == 2 && !one) { PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 2, true, false, 2>), grid, dim3(512), 0, st, a, groups); break; } \
        }                                                                                                                \
        if (one) PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 1, true>), grid, dim3(256), 0, st, a, groups);     \
        else PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 2, true>), grid, dim3(256), 0, st, a, groups);         \
        break;                                                                                                           \
      }                                                                                                                  \
    }                                                                                                                    \
    if constexpr (S >= 0 && NCT <= 2) {                                                                                  \
      if (a.x6) { PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, K, true, 2, false, true>), grid, dim3(256), 0, st, a, groups); break; } \
    }                                                                                                                    \
    if (a.bf16) { if (one && K == 1) PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 1>), grid, dim3(256), 0, st, a, groups);    \
                  else PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, K, true, 2>), grid, dim3(256), 0, st, a, groups); }            \
    else { if (one && K == 1) PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, false, 1>), grid, dim3(256), 0, st, a, groups);          \
           else PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, K, false, 2>), grid, dim3(256), 0, st, a, groups); }                  \
  } while (0)
  if (stem) { GO(-1, 1); return; }
  // slab count of the deepest-split input, as a compile-time loop bound of the summing loader: 2, 4 or 8 (a loader built for 8
  // slabs issues 8 clamped loads per fetch whatever the producer's split)
  const int km = a.ks0 > a.ks1 ? a.ks0 : a.ks1;
#define GOK(S) do { if (!slabs) GO(S, 1); else if (km <= 2) GO(S, 2); else if (km <= 4) GO(S, 4); else GO(S, 8); } while (0)
  if (a.mode0 == PSM_SRC_SAME) GOK(PSM_SRC_SAME);
  else if (a.mode0 == PSM_SRC_UPSAMPLE) {
    const bool seam_inside = a.c1 > 0 && (a.c0 % ((a.bf16 || a.x6) ? 32 : 16)) != 0;
    if (seam_inside) GOK(3);
    else GOK(PSM_SRC_UPSAMPLE);
  }
  else GOK(PSM_SRC_MAXPOOL);
#undef GOK
#undef GO
}

// arrangements 2-4 (round 6): finished bf16 inputs only
template <int TH, int WM, int NCT, int WN>
static hipError_t launch_variant_abf(const PsmConvArgs& a, dim3 grid, int groups, hipStream_t st) {
  const bool stem = (a.c0 % 4 != 0) || (a.c1 % 4 != 0);
  const bool seam_inside = a.mode0 == PSM_SRC_UPSAMPLE && a.c1 > 0 && (a.c0 % 32) != 0;
  if (stem || !a.bf16 || !a.in_bf || a.ks0 > 1 || a.ks1 > 1 || a.x6 || seam_inside) return hipErrorInvalidValue;
  const bool one = (a.n_chunks + a.ksplit - 1) / a.ksplit <= 1;
#define GOA(S)                                                                                                         \
  do {                                                                                                                 \
    if (one) PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 1, true>), grid, dim3(256), 0, st, a, groups);  \
    else PSM_LAUNCH((psm_conv3x3_kernel<TH, WM, NCT, WN, S, 1, true, 2, true>), grid, dim3(256), 0, st, a, groups);      \
  } while (0)
  if (a.mode0 == PSM_SRC_SAME) GOA(PSM_SRC_SAME);
  else if (a.mode0 == PSM_SRC_UPSAMPLE) GOA(PSM_SRC_UPSAMPLE);
  else GOA(PSM_SRC_MAXPOOL);
#undef GOA
  return hipSuccess;
}

hipError_t psm_launch_conv3x3(const PsmConvArgs& a, int arrangement, int nct, int n_cases, hipStream_t st) {
  const int cout_tiles = (a.cout + 15) / 16;
  const bool stem = (a.c0 % 4 != 0) || (a.c1 % 4 != 0);
  if (stem && (a.mode0 != PSM_SRC_SAME || a.c1 != 0 || a.ks0 != 1)) return hipErrorInvalidValue;
  if (a.ks0 > 8 || a.ks1 > 8 || a.c0 < 1 || (!stem && a.c0 < 4)) return hipErrorInvalidValue;
  if (a.mode0 != PSM_SRC_UPSAMPLE && a.c1 != 0) return hipErrorInvalidValue;     // skip inputs come with the upsample
  const int th = psm_conv_tile_rows(arrangement);
  if (th == 0 || nct != psm_conv_tile_nct(arrangement, nct)) return hipErrorInvalidValue;
  const int groups = (cout_tiles + nct - 1) / nct;
  const dim3 grid((a.W + TW - 1) / TW, (a.H + th - 1) / th, n_cases * groups * a.ksplit);
  switch (arrangement) {
    case 0:                        // pixel-major: 8 rows x 16 columns, every wave 2 rows x all NCT (1 or 2) channel tiles
      if (nct == 1) launch_variant<8, 2, 1, 1>(a, grid, groups, st); else launch_variant<8, 2, 2, 2>(a, grid, groups, st);
      break;
    case 1: launch_variant<2, 2, 4, 1>(a, grid, groups, st); break;    // channel-major: 2 rows x 16 columns, 4 waves = 4 channel tiles
    case 2: { const hipError_t e = launch_variant_abf<16, 4, 4, 4>(a, grid, groups, st); if (e != hipSuccess) return e; break; }
    case 3: { const hipError_t e = launch_variant_abf<16, 4, 2, 2>(a, grid, groups, st); if (e != hipSuccess) return e; break; }
    case 4: { const hipError_t e = launch_variant_abf<8, 2, 4, 4>(a, grid, groups, st); if (e != hipSuccess) return e; break; }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t psm_launch_conv_stem(const PsmConvArgs& a, int n_cases, hipStream_t st) {
  const int kg = (9 * a.c0 + 15) / 16;
  if (a.c0 < 1 || kg > 4 || a.cout > 16 || a.mode0 != PSM_SRC_SAME || a.c1 != 0 || a.ks0 != 1 || a.ksplit != 1) return hipErrorInvalidValue;
  const dim3 grid((a.W + TW - 1) / TW, (a.H + 7) / 8, n_cases);
  if (a.c0 == 3) PSM_LAUNCH((psm_conv_stem_kernel<2, 3>), grid, dim3(256), 0, st, a);          // (Ux, Uy, SDF)
  else if (a.c0 == 4) PSM_LAUNCH((psm_conv_stem_kernel<3, 4>), grid, dim3(256), 0, st, a);     // pressureSM_Poisson's 4 channels
  else if (kg == 1) PSM_LAUNCH((psm_conv_stem_kernel<1, 0>), grid, dim3(256), 0, st, a);
  else if (kg == 2) PSM_LAUNCH((psm_conv_stem_kernel<2, 0>), grid, dim3(256), 0, st, a);
  else if (kg == 3) PSM_LAUNCH((psm_conv_stem_kernel<3, 0>), grid, dim3(256), 0, st, a);
  else PSM_LAUNCH((psm_conv_stem_kernel<4, 0>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t psm_launch_head1x1(const PsmHeadArgs& a, hipStream_t st) {
  PSM_LAUNCH(psm_head1x1_kernel, dim3((unsigned)((a.n_pix + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}
