// psm_unet_api.cpp -- handle, layer schedule, weight packing and C-ABI of the convolutional path
// (include/psm_unet.h; kernels in psm_unet.hip).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "psm_alloc.h"

#include "../../include/psm.h"
#include "../../include/psm_unet.h"
#include "psm_unet.h"

namespace {
struct Conv {
  int k = 3, cin = 0, cout = 0, level = 0;
  int src = 0;          // 0 input image, 1 previous conv, 2 max-pool of previous conv, 3 upsample(previous) ++ skip
  int skip = -1;        // conv index whose output is concatenated (src == 3)
  int relu = 1;
  // launch configuration and packed operands
  int arrangement = 0, nct = 1, n_chunks = 0, groups = 1, ksplit = 1;
  int kw = 1;                  // 2: in-workgroup K split (PsmConvArgs::kw)
  bool stem = false;            // K = 9*c_in flattened (first layer on the raw image)
  bool fuse_head = false;       // this layer's epilogue also computes the 1x1 head
  bool in_bf = false, out_bf = false;   // bf16 mode: inputs read / output stored as bf16 (finished activations only)
  bool x6 = false;              // float32 mode: this layer runs the x6 form (three bf16 planes per operand, chunks of 32 channels; 8-row tiles only)
  // fused level pair (psm_unet_pair.hip): 1 = first convolution of a pair (its launch computes both), 2 = second (no launch)
  int pair = 0, pair_kind = 0;
  uint4* d_wpa = nullptr;       // pair leader: conv A's fragments
  uint4* d_wpb = nullptr;       // pair leader: conv B's fragments
  PsmPairTile* d_tiles = nullptr;   // pair leader: one descriptor per tile of the planned case batch (build_pair_tiles)
  std::vector<float> W, b;     // host copies (HWIO), kept for re-packing at plan time
  bool set = false;
  float4* d_w = nullptr;
  float* d_b = nullptr;
  float* d_w1 = nullptr;       // head: [cin][cout]
  float* d_out = nullptr;      // [ksplit][max_cases][H][W][cout] (ksplit > 1: partial-sum slabs, finished by the consumer's loader)
  int64_t slab = 0;
  // layout of d_out (act_layout): finished bf16 activations are zero-haloed, everything else is dense
  bool padded = false;
  int P = 0;                   // row pitch (pixels)
  int64_t case_elems = 0;      // elements per case
  int64_t origin = 0;          // element offset of pixel (0, 0) of case 0 within d_out
};
thread_local std::string g_err;
}  // namespace

struct psm_unet {
  int c_in = 0, c_out = 0, L = 0, device = 0;
  std::vector<int> widths;
  std::vector<Conv> convs;
  int ny = 0, nx = 0, max_cases = 0, last_cases = 0, bf16 = 0;
  bool planned = false;
  std::vector<int> tile_choice; // per convolution: -1 planner's tile, else 0 (8 rows x 2 channel tiles), 1 (8 x 1), 2 (2 rows x 4) -- psm_unet_autotune
  std::vector<int> pair_choice; // per convolution (pair leaders): -1 planner's rule (tile count), 0 never pair, 1 pair whenever a kernel exists
  std::vector<int> x6_choice;   // per convolution (float32 mode): -1 planner's rule (c_in >= 64), 0 float32 MFMA, 1 x6 wherever the kernel exists
  std::vector<int> ksplit_cap;  // per convolution: deepest split-K the planner may choose (psm_unet_autotune lowers it where a split does not pay)
  bool keep_act = false;        // fused pairs also store what they would keep on chip (introspection for the parity tests)
  bool x6 = true;               // float32 mode: 8-row-tile layers run on the bf16 matrix pipe with exactly split operands (PSM_UNET_X6=0: float32 MFMA)
  float *d_in = nullptr, *d_field = nullptr, *h_in = nullptr, *h_out = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
};

namespace {
int fail(psm_unet* u, int code, const std::string& m) { if (u) u->err = m; else g_err = m; return code; }
#define UCHK(u, expr)                                                                                   \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail((u), PSM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

void free_dev(void* p) { if (p) (void)psm_dev_free(p); }

// Finished bf16 activations (bf16 mode) live in ZERO-HALOED tensors: [case][ACT_PADT + H + ACT_PADB][ACT_PADL + W + ACT_PADR][C],
// the halo written once (memset at plan time) and never again -- every producer masks its stores to the image.  A consumer
// that stages whole tiles (the fused level pairs, psm_unet_pair.hip) then reads its halo -- the 'same' padding, the overhang of
// the last tile, a max-pool's doubled extent -- straight from memory: no clamps, no selects, and the tile can be moved by
// LDS-DMA, which cannot write zeros of its own.  The margins cover: 2 pixels of halo left / top at the tensor's own resolution
// and 4 when it is read through a 2x2 max-pool; right / bottom the last 30 x 14 tile's overhang + halo (32 / 16), doubled
// through a max-pool (64 / 32).  288 GB of HBM: the 20-45 % of extra footprint are never traffic.
constexpr int ACT_PADL = 4, ACT_PADT = 4, ACT_PADR = 64, ACT_PADB = 32;
void act_layout(Conv& c, int H, int W, bool padded) {
  c.padded = padded;
  if (padded) {
    c.P = ACT_PADL + W + ACT_PADR;
    c.case_elems = (int64_t)(ACT_PADT + H + ACT_PADB) * c.P * c.cout;
    c.origin = ((int64_t)ACT_PADT * c.P + ACT_PADL) * c.cout;
  } else { c.P = W; c.case_elems = (int64_t)H * W * c.cout; c.origin = 0; }
}
// pixel (0, 0) of case 0 of a convolution's output, as the kernels' float pointer (bf16 tensors are addressed in 2-byte elements)
float* act_ptr(const Conv& c) {
  return c.padded ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(c.d_out) + c.origin) : c.d_out;
}
// bytes of a zero-haloed bf16 tensor for the whole case batch: the pair kernels address it with 32-bit offsets
int64_t padded_bytes(int H, int W, int C, int cases) { return (int64_t)cases * (ACT_PADT + H + ACT_PADB) * (ACT_PADL + W + ACT_PADR) * C * 2; }

// Per-tile descriptors of a fused pair (PsmPairTile, psm_unet.h) for the planned case batch: tile t = (cs * tiles_y + by) * tiles_x + bx.
int build_pair_tiles(psm_unet* u, Conv& A, const Conv& B, const Conv* pv, const Conv* sk, int H, int W, int cases) {
  const int tx = (W + PSM_PAIR_TX - 1) / PSM_PAIR_TX, ty = (H + PSM_PAIR_TY - 1) / PSM_PAIR_TY;
  std::vector<PsmPairTile> t((size_t)cases * tx * ty);
  for (int cs = 0; cs < cases; ++cs)
    for (int by = 0; by < ty; ++by)
      for (int bx = 0; bx < tx; ++bx) {
        PsmPairTile& d = t[((size_t)cs * ty + by) * tx + bx];
        const int y0 = by * PSM_PAIR_TY, x0 = bx * PSM_PAIR_TX;
        d.y0 = y0; d.x0 = x0; d.cs = cs;
        d.pix = (cs * H + y0) * W + x0;
        d.flags = (x0 >= 2 && x0 + 33 <= W && y0 >= 2 && y0 + 16 <= H) ? 1 : 0;
        d.offo = (int)(((int64_t)cs * B.case_elems + ((int64_t)y0 * B.P + x0) * B.cout) * 2);
        d.off0 = d.off1 = 0;
        if (A.pair_kind == PSM_PAIR_UPCAT) {
          d.off0 = (int)(((int64_t)cs * pv->case_elems + ((int64_t)(y0 / 2 - 1) * pv->P + (x0 / 2 - 1)) * pv->cout) * 2);
          d.off1 = (int)(((int64_t)cs * sk->case_elems + ((int64_t)(y0 - 2) * sk->P + (x0 - 2)) * sk->cout) * 2);
        } else if (A.pair_kind == PSM_PAIR_POOL) {
          d.off0 = (int)(((int64_t)cs * pv->case_elems + ((int64_t)(2 * y0 - 4) * pv->P + (2 * x0 - 4)) * pv->cout) * 2);
        }
      }
  free_dev(A.d_tiles); A.d_tiles = nullptr;
  UCHK(u, psm_dev_malloc((void**)&A.d_tiles, t.size() * sizeof(PsmPairTile)));
  UCHK(u, psm_copy_h2d(A.d_tiles, t.data(), t.size() * sizeof(PsmPairTile)));
  return PSM_OK;
}

// MFMA operand order: wpack[cog][chunk g][tap][ct][lane][j] = W[tap][16g + 4*(lane>>4) + j][16*(cog*nct + ct) + (lane&15)]
std::vector<float> pack_conv3x3(const Conv& c) {
  const int chunks = c.n_chunks, nct = c.nct, groups = c.groups;
  std::vector<float> p((size_t)groups * chunks * 9 * nct * 64 * 4, 0.f);
  for (int cog = 0; cog < groups; ++cog)
    for (int g = 0; g < chunks; ++g)
      for (int tap = 0; tap < 9; ++tap)
        for (int ct = 0; ct < nct; ++ct)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int ci = 16 * g + 4 * (lane >> 4) + j, co = 16 * (cog * nct + ct) + (lane & 15);
              if (ci < c.cin && co < c.cout)
                p[(((((size_t)cog * chunks + g) * 9 + tap) * nct + ct) * 64 + lane) * 4 + j] =
                    c.W[((size_t)tap * c.cin + ci) * c.cout + co];
            }
  return p;
}

uint16_t f2bf(float f) {       // round to nearest even, like v_cvt_pk_bf16_f32
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// bf16: wpack[cog][chunk g of 32][tap][ct][lane][j < 8] = bf16(W[tap][32g + 8*(lane>>4) + j][16*(cog*nct + ct) + (lane&15)])
std::vector<uint16_t> pack_conv3x3_bf16(const Conv& c) {
  const int chunks = c.n_chunks, nct = c.nct, groups = c.groups;
  std::vector<uint16_t> p((size_t)groups * chunks * 9 * nct * 64 * 8, 0);
  for (int cog = 0; cog < groups; ++cog)
    for (int g = 0; g < chunks; ++g)
      for (int tap = 0; tap < 9; ++tap)
        for (int ct = 0; ct < nct; ++ct)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int ci = 32 * g + 8 * (lane >> 4) + j, co = 16 * (cog * nct + ct) + (lane & 15);
              if (ci < c.cin && co < c.cout)
                p[(((((size_t)cog * chunks + g) * 9 + tap) * nct + ct) * 64 + lane) * 8 + j] =
                    f2bf(c.W[((size_t)tap * c.cin + ci) * c.cout + co]);
            }
  return p;
}

// x6 (float32 mode on the bf16 matrix pipe): every weight split exactly into hi + mid + lo bf16 planes;
// wpack[cog][chunk g of 32][plane][tap][ct][lane][j < 8] = plane(W[tap][32g + 8*(lane>>4) + j][16*(cog*nct + ct) + (lane&15)])
float bf2f(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
std::vector<uint16_t> pack_conv3x3_x6(const Conv& c) {
  const int chunks = c.n_chunks, nct = c.nct, groups = c.groups;
  std::vector<uint16_t> p((size_t)groups * chunks * 3 * 9 * nct * 64 * 8, 0);
  for (int cog = 0; cog < groups; ++cog)
    for (int g = 0; g < chunks; ++g)
      for (int tap = 0; tap < 9; ++tap)
        for (int ct = 0; ct < nct; ++ct)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int ci = 32 * g + 8 * (lane >> 4) + j, co = 16 * (cog * nct + ct) + (lane & 15);
              if (ci >= c.cin || co >= c.cout) continue;
              const float w = c.W[((size_t)tap * c.cin + ci) * c.cout + co];
              const uint16_t h = f2bf(w);
              const float r1 = w - bf2f(h);                     // exact
              const uint16_t m = f2bf(r1);
              const float r2 = r1 - bf2f(m);                    // exact
              const uint16_t pl[3] = {h, m, f2bf(r2)};
              for (int q = 0; q < 3; ++q)
                p[((((((size_t)cog * chunks + g) * 3 + q) * 9 + tap) * nct + ct) * 64 + lane) * 8 + j] = pl[q];
            }
  return p;
}

// stem: wpack[g][lane][j] = W[k = 16g + 4*(lane>>4) + j -> (tap = k / c_in, ci = k % c_in)][co = lane & 15]
std::vector<float> pack_stem(const Conv& c, bool bf16) {
  const int K = 9 * c.cin, kg = (K + 15) / 16;
  std::vector<float> p((size_t)kg * 64 * 4, 0.f);
  for (int g = 0; g < kg; ++g)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * g + 4 * (lane >> 4) + j, co = lane & 15;
        if (k < K && co < c.cout) {
          float w = c.W[((size_t)(k / c.cin) * c.cin + (k % c.cin)) * c.cout + co];
          if (bf16) { const uint32_t u = (uint32_t)f2bf(w) << 16; std::memcpy(&w, &u, 4); }
          p[((size_t)g * 64 + lane) * 4 + j] = w;
        }
      }
  return p;
}

// Fragments of the fused-pair kernels (psm_unet_pair.hip): 64 lanes x 8 bf16 each, the MFMA's FIRST operand: lane l holds
// W[tap][ci(l >> 4, j)][co = 16*nt + (l & 15)], j < 8.
//   stem (flat):        step s:            k = 32 s + 8 (l >> 4) + j -> (tap = k / c_in, ci = k % c_in), zero beyond 9 c_in
//   32-channel chunk:   [kx][ky][nt]:      ci = base + 8 (l >> 4) + j
//   16-channel chunk:   [pair s][ky][nt]:  kx = 2 s + (l >> 5), ci = base + 8 ((l >> 4) & 1) + j, zero for kx = 3
// Chunks: the channels of in0 then of in1 (c0 + c1 = c.cin), 32 at a time while they last, then one of 16.
std::vector<uint16_t> pack_pair(const Conv& c, bool flat, int c0, int c1) {
  const int NT = c.cout / 16;
  std::vector<uint16_t> p;
  auto frag = [&](auto&& wat) {                       // wat(kq, j, co) -> weight
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) p.push_back(f2bf(wat(lane >> 4, j, lane & 15)));
  };
  auto W = [&](int tap, int ci, int co) { return c.W[((size_t)tap * c.cin + ci) * c.cout + co]; };
  if (flat) {
    const int K = 9 * c.cin;
    for (int s = 0; s < (K + 31) / 32; ++s)
      frag([&](int kq, int j, int co) { const int k = 32 * s + 8 * kq + j; return k < K ? W(k / c.cin, k % c.cin, co) : 0.f; });
    return p;
  }
  for (int part = 0; part < 2; ++part) {
    const int cn = part == 0 ? c0 : c1, base = part == 0 ? 0 : c0;
    for (int cb = 0; cb < cn; cb += 32) {
      if (cn - cb >= 32) {
        for (int kx = 0; kx < 3; ++kx) for (int ky = 0; ky < 3; ++ky) for (int nt = 0; nt < NT; ++nt)
          frag([&](int kq, int j, int co) { return W(ky * 3 + kx, base + cb + 8 * kq + j, 16 * nt + co); });
      } else {
        for (int s = 0; s < 2; ++s) for (int ky = 0; ky < 3; ++ky) for (int nt = 0; nt < NT; ++nt)
          frag([&](int kq, int j, int co) { const int kx = 2 * s + (kq >> 1); return kx < 3 ? W(ky * 3 + kx, base + cb + 8 * (kq & 1) + j, 16 * nt + co) : 0.f; });
      }
    }
  }
  return p;
}

// workgroup count first (fill 256 CUs), then the most reuse per workgroup
// forced: index into the candidate list (psm_unet_autotune's measured tile choice), -1 = by rule; the split rule below then
// runs for the tile that was chosen, not for the rule's.
// big_ok (round 6): the layer may take arrangement 2 (16 rows x 64 channels per workgroup, a 64 px x 64 channel register block per
// wave: half the LDS operand reads per MFMA and half the staged bytes per flop) -- bf16 mode, not the stem; whether its inputs
// really are finished bf16 activations is only known after every layer's split is (psm_unet_plan falls back to the 8-row tile
// otherwise).  Rule from profiles/r06_conv_experiments.txt (2): only where such tiles still fill the chip AND the layer walks six or
// more channel chunks (64 cases of 256 x 256: dec3a 72.4 -> 65.4 us, enc4b 27.6 -> 24.4, dec2a 82.3 -> 78.5; shorter layers and
// every layer at 8 cases per step are slower with it).
void choose_config(Conv& c, int H, int W, int n_cases, bool can_split, int chunk_ch, int ks_cap = 8, bool x6_ok = false, int forced = -1,
                   bool big_ok = false, bool in_split = false) {
  const int ctiles = (c.cout + 15) / 16;
  struct Cand { int arr, nct, th; };
  const Cand cands[4] = {{0, 2, 8}, {0, 1, 8}, {1, 4, 2}, {2, 4, 16}};
  // diagnostic knobs (tools/attic/unet_bench.py sweeps): workgroups wanted before reuse counts, deepest split
  const long fill = getenv("PSM_UNET_FILL") ? atol(getenv("PSM_UNET_FILL")) : 256;
  const int ks_max = getenv("PSM_UNET_KSPLIT_MAX") ? atoi(getenv("PSM_UNET_KSPLIT_MAX")) : 8;
  long best_score = -1;
  const bool use_forced = forced >= 0 && forced < (big_ok ? 4 : 3) && cands[forced].nct <= ctiles;
  if (use_forced) {
    const Cand& k = cands[forced];
    c.arrangement = k.arr; c.nct = k.nct; c.groups = (ctiles + k.nct - 1) / k.nct;
  }
  for (int ki = 0; ki < 3 && !use_forced; ++ki) {
    const Cand& k = cands[ki];
    if (k.nct > ctiles) continue;                       // never compute padded channel tiles
    const int groups = (ctiles + k.nct - 1) / k.nct;
    const long wgs = (long)((W + 15) / 16) * ((H + k.th - 1) / k.th) * groups * n_cases;
    const long reuse = (long)k.nct * k.th;
    // equal workgroup count / reuse: the 2-row x 64-channel tile (arrangement 1), except for bf16 layers of four or more 32-channel chunks, where the
    // 8-row x 16-channel tile stages half the bytes per chunk for the same MFMAs and can take the in-workgroup K split (psm_unet_plan): enc4b at 8 cases
    // 8.9 us as 2 x 64, 7.7 us as 8 x 16, 6.8 us with the split on top; 512 x 512 x 1: enc3b 5.6 -> 5.0, dec3b 5.4 -> 4.6 us (profiles/r06_conv_experiments.txt (8))
    const bool kw_off = getenv("PSM_UNET_KW") && atoi(getenv("PSM_UNET_KW")) == 0;              // PSM_UNET_KW=0: the planner without either change
    // (in_split: an input arrives as float32 partial-sum slabs -- the summing loader, no in-workgroup split: 512 x 512 x 1, enc4b behind a split enc4a, 10.1 us as 2 x 64, 11.2 us as 8 x 16)
    const bool long_bf16 = chunk_ch == 32 && !x6_ok && !kw_off && !in_split && (c.cin + chunk_ch - 1) / chunk_ch >= 4;
    const long tie = long_bf16 ? (k.arr == 0 && k.nct == 1 ? 1 : 0) : (k.arr ? 1 : 0);
    const long score = wgs >= fill ? 1000000 + reuse * 1000 + tie : wgs * 10 + tie;
    if (score > best_score) { best_score = score; c.arrangement = k.arr; c.nct = k.nct; c.groups = groups; }
  }
  if (!use_forced && big_ok && ctiles >= 4 && ctiles % 4 == 0 && (c.cin + chunk_ch - 1) / chunk_ch >= 6 && getenv("PSM_UNET_NO_BIG_TILES") == nullptr) {
    const long wgs16 = (long)((W + 15) / 16) * ((H + 15) / 16) * (ctiles / 4) * n_cases;
    if (wgs16 >= fill) { c.arrangement = 2; c.nct = 4; c.groups = ctiles / 4; }
  }
  c.x6 = x6_ok && c.arrangement == 0;                 // the x6 form exists for the 8-row tiles (psm_unet.hip)
  if (c.x6) chunk_ch = 32;
  c.n_chunks = (c.cin + chunk_ch - 1) / chunk_ch;
  // split the input channels over workgroups until the chip is filled (partial-sum slabs, see psm_unet.h); only
  // layers whose output feeds another convolution can be split, at most 8 ways
  const int th = psm_conv_tile_rows(c.arrangement);
  const long wgs = (long)((W + 15) / 16) * ((H + th - 1) / th) * c.groups * n_cases;
  c.ksplit = 1;
  // chunks per split, at least: one for layers of four or more chunks, two below.  Measured at batch 1 against "two
  // everywhere": float32 174 -> 167 us, bf16 107 -> 98 us (the 16^2 layer with 4 chunks runs as 128 workgroups instead of
  // 64); one everywhere also splits the 2-chunk layers, whose consumers then read float32 slabs: bf16 105 us.  No change at
  // 8 cases per step.
  const int min_chunks = getenv("PSM_UNET_SPLIT_MIN_CHUNKS") ? atoi(getenv("PSM_UNET_SPLIT_MIN_CHUNKS")) : (c.n_chunks >= 4 ? 1 : 2);
  // Whether a split pays is not decidable from the workgroup count alone: 512 x 512 batch 1 bf16, enc4b split in two runs in
  // 9.4 us against 9.2 us whole while its consumer dec3a then reads float32 partial-sum slabs through the summing loader
  // instead of finished bf16 activations (25.6 against 13.4 us); 256 x 256 batch 1, dec2a split in two: 9.0 against 14.3 us.
  // psm_unet_autotune measures it per layer (ksplit_cap); this is the starting point.
  // (A bf16 layer that can take the IN-WORKGROUP split -- 8-row tile, four or more chunks, at most one workgroup per CU, inputs that are not slabs themselves -- is
  // not split over workgroups: its consumer keeps reading finished bf16 activations.  512 x 512 x 1: enc4a / enc4b / dec3a / dec3b 5.2 + 7.0 + 16.9 + 5.3 us with the
  // splits over workgroups the rule below starts from (the autotuner got them to 4.3 + 10.1 + 8.5 + 6.5), 6.3 + 6.4 + 8.5 + 5.3 us this way.)
  const bool kw_off = getenv("PSM_UNET_KW") && atoi(getenv("PSM_UNET_KW")) == 0;
  const bool kw_ok = chunk_ch == 32 && !c.x6 && !kw_off && !in_split && c.arrangement == 0 && c.n_chunks >= 4 && wgs <= 256;
  while (can_split && !kw_ok && wgs * c.ksplit < fill && c.ksplit < std::min(ks_max, ks_cap) && c.n_chunks / (c.ksplit * 2) >= min_chunks) c.ksplit *= 2;
}

int upload_conv(psm_unet* u, Conv& c) {
  free_dev(c.d_w); c.d_w = nullptr; free_dev(c.d_b); c.d_b = nullptr; free_dev(c.d_w1); c.d_w1 = nullptr;
  free_dev(c.d_wpa); c.d_wpa = nullptr; free_dev(c.d_wpb); c.d_wpb = nullptr; free_dev(c.d_tiles); c.d_tiles = nullptr;
  std::vector<float> bias((size_t)((c.cout + 15) / 16 + 4) * 16, 0.f);
  std::memcpy(bias.data(), c.b.data(), c.cout * sizeof(float));
  UCHK(u, psm_dev_malloc((void**)&c.d_b, bias.size() * sizeof(float)));
  UCHK(u, psm_copy_h2d(c.d_b, bias.data(), bias.size() * sizeof(float)));
  if (c.k == 3 && c.stem) {
    const std::vector<float> p = pack_stem(c, u->bf16 != 0);
    UCHK(u, psm_dev_malloc((void**)&c.d_w, p.size() * sizeof(float)));
    UCHK(u, psm_copy_h2d(c.d_w, p.data(), p.size() * sizeof(float)));
  } else if (c.k == 3 && c.x6) {
    const std::vector<uint16_t> p = pack_conv3x3_x6(c);
    UCHK(u, psm_dev_malloc((void**)&c.d_w, p.size() * sizeof(uint16_t)));
    UCHK(u, psm_copy_h2d(c.d_w, p.data(), p.size() * sizeof(uint16_t)));
  } else if (c.k == 3 && u->bf16) {
    const std::vector<uint16_t> p = pack_conv3x3_bf16(c);
    UCHK(u, psm_dev_malloc((void**)&c.d_w, p.size() * sizeof(uint16_t)));
    UCHK(u, psm_copy_h2d(c.d_w, p.data(), p.size() * sizeof(uint16_t)));
  } else if (c.k == 3) {
    const std::vector<float> p = pack_conv3x3(c);
    UCHK(u, psm_dev_malloc((void**)&c.d_w, p.size() * sizeof(float)));
    UCHK(u, psm_copy_h2d(c.d_w, p.data(), p.size() * sizeof(float)));
  } else {
    UCHK(u, psm_dev_malloc((void**)&c.d_w1, c.W.size() * sizeof(float)));
    UCHK(u, psm_copy_h2d(c.d_w1, c.W.data(), c.W.size() * sizeof(float)));
  }
  if (c.pair == 1) {
    const size_t ci = &c - u->convs.data();
    const Conv& B = u->convs[ci + 1];
    int c0 = c.cin, c1 = 0;
    if (c.pair_kind == PSM_PAIR_UPCAT) { c0 = u->convs[ci - 1].cout; c1 = u->convs[c.skip].cout; }
    const std::vector<uint16_t> pa = pack_pair(c, c.pair_kind == PSM_PAIR_STEM, c0, c1), pb = pack_pair(B, false, B.cin, 0);
    UCHK(u, psm_dev_malloc((void**)&c.d_wpa, pa.size() * sizeof(uint16_t)));
    UCHK(u, psm_copy_h2d(c.d_wpa, pa.data(), pa.size() * sizeof(uint16_t)));
    UCHK(u, psm_dev_malloc((void**)&c.d_wpb, pb.size() * sizeof(uint16_t)));
    UCHK(u, psm_copy_h2d(c.d_wpb, pb.data(), pb.size() * sizeof(uint16_t)));
  }
  return PSM_OK;
}

int forward(psm_unet* u, const float* d_grid, int n, float* d_field, hipStream_t st, hipEvent_t* ev = nullptr, int n_layers = -1) {
  const size_t n_run = n_layers < 0 ? u->convs.size() : (size_t)n_layers;
  for (size_t i = 0; i < n_run; ++i) {
    if (ev) UCHK(u, hipEventRecord(ev[i], st));
    if (psm_launch_probe) psm_launch_probe->tag = (int)i;
    Conv& c = u->convs[i];
    const int H = u->ny >> c.level, W = u->nx >> c.level;
    float* out = (i + 1 == u->convs.size()) ? d_field : act_ptr(c);
    if (c.pair == 2) continue;                                // computed by the pair's first convolution's launch
    if (c.pair == 1) {
      const Conv& B = u->convs[i + 1];
      PsmPairArgs p{};
      p.wA = c.d_wpa; p.wB = c.d_wpb; p.biasA = c.d_b; p.biasB = B.d_b;
      p.H = H; p.W = W; p.tiles_x = (W + PSM_PAIR_TX - 1) / PSM_PAIR_TX; p.tiles_y = (H + PSM_PAIR_TY - 1) / PSM_PAIR_TY; p.n_cases = n;
      p.out_case = B.case_elems; p.PO = B.P;                // conv A's kept activation (mid_out) has the same layout
      p.tiles = c.d_tiles; p.cm = c.cout;
      p.out = (B.fuse_head && !u->keep_act) ? nullptr : reinterpret_cast<unsigned short*>(act_ptr(B));
      p.mid_out = u->keep_act ? reinterpret_cast<unsigned short*>(act_ptr(c)) : nullptr;
      if (c.pair_kind == PSM_PAIR_STEM) { p.in0 = d_grid; p.c0 = c.cin; p.in0_case = (int64_t)H * W * c.cin; p.P0 = W; }
      else {
        const Conv& pv = u->convs[i - 1];
        p.in0 = act_ptr(pv); p.c0 = pv.cout; p.in0_case = pv.case_elems; p.P0 = pv.P;
        if (c.pair_kind == PSM_PAIR_UPCAT) { const Conv& sk = u->convs[c.skip]; p.in1 = act_ptr(sk); p.c1 = sk.cout; p.in1_case = sk.case_elems; p.P1 = sk.P; }
      }
      if (B.fuse_head) {
        const Conv& hd = u->convs[i + 2];
        p.head_w = hd.d_w1; p.head_b = hd.d_b; p.head_out = d_field; p.head_cout = hd.cout; p.head_case = (int64_t)H * W * hd.cout;
      }
      UCHK(u, psm_launch_conv_pair(p, c.pair_kind, c.cout, n, st));
      continue;
    }
    if (c.k == 1) {
      if (u->convs[i - 1].fuse_head) continue;               // computed in the previous layer's epilogue
      PsmHeadArgs ha{u->convs[i - 1].d_out, c.d_w1, c.d_b, out, (int64_t)n * H * W, c.cin, c.cout};     // its input is float32 (dense) by the planner's rule
      UCHK(u, psm_launch_head1x1(ha, st));
      continue;
    }
    PsmConvArgs a{};
    a.wpack = c.d_w; a.bias = c.d_b; a.out = out; a.n_chunks = c.n_chunks; a.H = H; a.W = W; a.cout = c.cout; a.relu = c.relu;
    a.out_case = (i + 1 == u->convs.size()) ? (int64_t)H * W * c.cout : c.case_elems;
    a.PO = (i + 1 == u->convs.size()) ? W : c.P;
    a.kw = c.kw; a.ksplit = c.ksplit; a.out_slab = c.slab; a.ks0 = 1; a.ks1 = 1; a.bf16 = u->bf16;
    a.in_bf = c.in_bf ? 1 : 0; a.out_bf = c.out_bf ? 1 : 0; a.x6 = c.x6 ? 1 : 0;
    if (c.src == 0) { a.in0 = d_grid; a.c0 = c.cin; a.mode0 = PSM_SRC_SAME; a.H0 = H; a.W0 = W; a.P0 = W; a.in0_case = (int64_t)H * W * c.cin; }
    else {
      const Conv& pv = u->convs[i - 1];
      a.in0 = act_ptr(pv); a.c0 = pv.cout; a.ks0 = pv.ksplit; a.slab0 = pv.slab; a.pbias0 = pv.d_b; a.P0 = pv.P; a.in0_case = pv.case_elems;
      if (c.src == 1) { a.mode0 = PSM_SRC_SAME; a.H0 = H; a.W0 = W; }
      else if (c.src == 2) { a.mode0 = PSM_SRC_MAXPOOL; a.H0 = 2 * H; a.W0 = 2 * W; }
      else { const Conv& sk = u->convs[c.skip];
             a.mode0 = PSM_SRC_UPSAMPLE; a.H0 = H / 2; a.W0 = W / 2; a.in1 = act_ptr(sk); a.c1 = sk.cout;
             a.ks1 = sk.ksplit; a.slab1 = sk.slab; a.pbias1 = sk.d_b; a.in1_case = sk.case_elems; a.P1 = sk.P; }
    }
    if (c.fuse_head) {
      const Conv& hd = u->convs[i + 1];
      a.head_w = hd.d_w1; a.head_b = hd.d_b; a.head_out = d_field; a.head_cout = hd.cout; a.head_case = (int64_t)H * W * hd.cout;
    }
    if (c.stem) UCHK(u, psm_launch_conv_stem(a, n, st));
    else UCHK(u, psm_launch_conv3x3(a, c.arrangement, c.nct, n, st));
  }
  if (ev) UCHK(u, hipEventRecord(ev[u->convs.size()], st));
  u->last_cases = n;
  return PSM_OK;
}
}  // namespace

extern "C" {

const char* psm_unet_last_error(const psm_unet* u) { return u ? u->err.c_str() : g_err.c_str(); }

int psm_unet_create(int32_t c_in, int32_t c_out, int32_t n_levels, const int32_t* widths, int32_t device, psm_unet** out) {
  if (!out || !widths) return fail(nullptr, PSM_ERR_ARG, "null argument");
  *out = nullptr;
  if (c_in < 1 || c_in > 16 || c_out < 1 || c_out > 16) return fail(nullptr, PSM_ERR_ARG, "c_in / c_out must be 1..16");
  if (n_levels < 2 || n_levels > 7) return fail(nullptr, PSM_ERR_ARG, "n_levels must be 2..7");
  for (int l = 0; l < n_levels; ++l)
    if (widths[l] < 16 || widths[l] > 1024 || widths[l] % 16) return fail(nullptr, PSM_ERR_ARG, "widths must be multiples of 16 in [16, 1024]");
  if (widths[0] > 64) return fail(nullptr, PSM_ERR_ARG, "first-level width above 64 is not supported by the 1x1 head");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, PSM_ERR_NO_DEVICE, "no HIP device: the convolutional path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, PSM_ERR_ARG, "device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, PSM_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, PSM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  psm_unet* u = new psm_unet();
  u->c_in = c_in; u->c_out = c_out; u->L = n_levels; u->device = device;
  u->widths.assign(widths, widths + n_levels);
  std::vector<int> enc_last(n_levels);
  for (int l = 0; l < n_levels; ++l) {
    Conv a; a.cin = l == 0 ? c_in : widths[l - 1]; a.cout = widths[l]; a.level = l; a.src = l == 0 ? 0 : 2;
    Conv b; b.cin = widths[l]; b.cout = widths[l]; b.level = l; b.src = 1;
    u->convs.push_back(a); u->convs.push_back(b);
    enc_last[l] = (int)u->convs.size() - 1;
  }
  for (int l = n_levels - 2; l >= 0; --l) {
    Conv a; a.cin = widths[l + 1] + widths[l]; a.cout = widths[l]; a.level = l; a.src = 3; a.skip = enc_last[l];
    Conv b; b.cin = widths[l]; b.cout = widths[l]; b.level = l; b.src = 1;
    u->convs.push_back(a); u->convs.push_back(b);
  }
  Conv h; h.k = 1; h.cin = widths[0]; h.cout = c_out; h.level = 0; h.src = 1; h.relu = 0;
  u->convs.push_back(h);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking) != hipSuccess) {
    delete u;
    return fail(nullptr, PSM_ERR_HIP, "cannot create a stream on the device");
  }
  *out = u;
  return PSM_OK;
}

void psm_unet_destroy(psm_unet* u) {
  if (!u) return;
  (void)hipSetDevice(u->device);
  if (u->stream) (void)hipStreamSynchronize(u->stream);
  for (Conv& c : u->convs) { free_dev(c.d_w); free_dev(c.d_b); free_dev(c.d_w1); free_dev(c.d_out); free_dev(c.d_wpa); free_dev(c.d_wpb); free_dev(c.d_tiles); }
  free_dev(u->d_in); free_dev(u->d_field);
  if (u->h_in) (void)hipHostFree(u->h_in);
  if (u->h_out) (void)hipHostFree(u->h_out);
  if (u->stream) (void)hipStreamDestroy(u->stream);
  delete u;
}

int psm_unet_num_convs(const psm_unet* u) { return u ? (int)u->convs.size() : PSM_ERR_ARG; }

int psm_unet_conv_shape(const psm_unet* u, int32_t idx, int32_t* k, int32_t* c_in, int32_t* c_out) {
  if (!u || idx < 0 || idx >= (int)u->convs.size() || !k || !c_in || !c_out) return PSM_ERR_ARG;
  *k = u->convs[idx].k; *c_in = u->convs[idx].cin; *c_out = u->convs[idx].cout;
  return PSM_OK;
}

int psm_unet_set_conv(psm_unet* u, int32_t idx, const float* weight, const float* bias) {
  if (!u) return PSM_ERR_ARG;
  if (idx < 0 || idx >= (int)u->convs.size()) return fail(u, PSM_ERR_ARG, "convolution index out of range");
  if (!weight || !bias) return fail(u, PSM_ERR_ARG, "null weights");
  Conv& c = u->convs[idx];
  c.W.assign(weight, weight + (size_t)c.k * c.k * c.cin * c.cout);
  c.b.assign(bias, bias + c.cout);
  c.set = true;
  if (u->planned) { UCHK(u, hipSetDevice(u->device)); return upload_conv(u, c); }
  return PSM_OK;
}

int psm_unet_set_precision(psm_unet* u, int32_t precision) {
  if (!u) return PSM_ERR_ARG;
  if (precision != PSM_PRECISION_F32 && precision != PSM_PRECISION_BF16) return fail(u, PSM_ERR_ARG, "unknown precision");
  if (u->planned) return fail(u, PSM_ERR_STATE, "set the precision before psm_unet_plan");
  u->bf16 = precision == PSM_PRECISION_BF16 ? 1 : 0;
  return PSM_OK;
}

int psm_unet_keep_activations(psm_unet* u, int32_t on) {
  if (!u) return PSM_ERR_ARG;
  if (u->planned) return fail(u, PSM_ERR_STATE, "call psm_unet_keep_activations before psm_unet_plan");
  u->keep_act = on != 0;
  return PSM_OK;
}

int psm_unet_plan(psm_unet* u, int32_t ny, int32_t nx, int32_t max_cases) {
  if (!u) return PSM_ERR_ARG;
  for (const Conv& c : u->convs) if (!c.set) return fail(u, PSM_ERR_STATE, "model incomplete: call psm_unet_set_conv for every convolution first");
  const int m = 1 << (u->L - 1);
  if (ny < m || nx < m || ny % m || nx % m) return fail(u, PSM_ERR_ARG, "ny and nx must be multiples of 2^(n_levels-1)");
  if (max_cases < 1 || (int64_t)ny * nx * max_cases > ((int64_t)1 << 28)) return fail(u, PSM_ERR_ARG, "bad case batch");
  UCHK(u, hipSetDevice(u->device));
  UCHK(u, hipStreamSynchronize(u->stream));
  u->ny = ny; u->nx = nx; u->max_cases = max_cases;
  u->x6 = !(getenv("PSM_UNET_X6") && atoi(getenv("PSM_UNET_X6")) == 0);
  for (int pass = 0; pass < 2; ++pass) {
  if (pass == 1) {
    // bf16 activation storage (bf16 mode): a finished activation is stored as bf16 when every consumer reads bf16, i.e. is
    // a 3x3 layer none of whose inputs arrives as split-K slabs (those loaders sum float32 slabs) and whose concatenation
    // seam lies on a chunk boundary; a consumer with one float32 input takes all its inputs in float32.
    const size_t n = u->convs.size();
    // level pairs come first: two convolutions that the pair kernels could take (shape and tile count) are not split over K
    // -- a pair has no slabs, and a level large enough for a pair fills the chip without them
    const long pair_min = getenv("PSM_UNET_PAIR_MIN") ? atol(getenv("PSM_UNET_PAIR_MIN")) : 96;
    for (size_t i = 0; u->bf16 && getenv("PSM_UNET_NO_PAIR") == nullptr && i + 1 < n; ++i) {
      Conv& A = u->convs[i]; Conv& B = u->convs[i + 1];
      if (A.k != 3 || B.k != 3 || B.src != 1 || A.level != B.level || B.cin != A.cout || B.cout != A.cout || (A.cout != 16 && A.cout != 32)) continue;
      const int H = ny >> A.level, W = nx >> A.level;
      const int pc = i < u->pair_choice.size() ? u->pair_choice[i] : -1;
      if (pc == 0 || (pc < 0 && (long)((W + PSM_PAIR_TX - 1) / PSM_PAIR_TX) * ((H + PSM_PAIR_TY - 1) / PSM_PAIR_TY) * max_cases < pair_min)) continue;
      A.ksplit = 1; B.ksplit = 1;
    }
    std::vector<char> obf(n, 0);
    for (size_t i = 0; i < n; ++i) obf[i] = (u->bf16 && u->convs[i].k == 3 && u->convs[i].ksplit == 1 && getenv("PSM_UNET_F32_ACT") == nullptr) ? 1 : 0;
    for (bool changed = true; changed;) {
      changed = false;
      for (size_t i = 0; i < n; ++i) {
        Conv& c = u->convs[i];
        std::vector<int> ins;
        if (c.src != 0) ins.push_back((int)i - 1);
        if (c.src == 3) ins.push_back(c.skip);
        bool ok = c.k == 3 && !c.stem && c.src != 0;
        for (int j : ins) ok = ok && obf[j] && u->convs[j].ksplit == 1;
        if (c.src == 3 && (u->convs[i - 1].cout % 32) != 0) ok = false;            // seam inside a 32-channel chunk
        if (c.k == 1 && i > 0 && u->convs[i - 1].fuse_head) continue;              // fused head reads registers
        c.in_bf = ok;
        if (!ok) for (int j : ins) if (obf[j]) { obf[j] = 0; changed = true; }
      }
    }
    for (size_t i = 0; i < n; ++i) u->convs[i].out_bf = obf[i] != 0;
    // arrangements 2-4 exist for finished bf16 inputs only (psm_unet.hip, launch_variant_abf): back to the 8-row tile otherwise
    for (Conv& c : u->convs)
      if (c.k == 3 && c.arrangement >= 2 && !c.in_bf) { c.arrangement = 0; c.nct = 2; c.groups = ((c.cout + 15) / 16 + 1) / 2; }
    // fused level pairs (psm_unet_pair.hip): both 3x3 convolutions of a level in one launch, where the level is wide
    // enough to fill the chip with 30 x 14 tiles and the shapes are ones the pair kernels are written for
    for (size_t i = 0; i + 1 < n; ++i) { u->convs[i].pair = 0; u->convs[i + 1].pair = 0; }
    for (size_t i = 0; u->bf16 && getenv("PSM_UNET_NO_PAIR") == nullptr && i + 1 < n; ++i) {
      Conv& A = u->convs[i]; Conv& B = u->convs[i + 1];
      if (A.k != 3 || B.k != 3 || B.src != 1 || A.level != B.level || A.pair || B.pair) continue;
      const int cm = A.cout, H = ny >> A.level, W = nx >> A.level;
      if (B.cin != cm || B.cout != cm || (cm != 16 && cm != 32) || A.ksplit != 1 || B.ksplit != 1) continue;
      if (!(B.out_bf || B.fuse_head)) continue;
      const long wgs = (long)((W + PSM_PAIR_TX - 1) / PSM_PAIR_TX) * ((H + PSM_PAIR_TY - 1) / PSM_PAIR_TY) * max_cases;
      const int pc = i < u->pair_choice.size() ? u->pair_choice[i] : -1;
      if (pc == 0 || (pc < 0 && wgs < pair_min)) continue;
      // the source transform names the kind; whether a kernel exists for these channel counts (and a fused head) is the
      // launcher's own predicate
      int kind = -1, c0 = A.cin, c1 = 0;
      if (A.src == 0) kind = PSM_PAIR_STEM;
      else if (A.src == 2 && A.in_bf) kind = PSM_PAIR_POOL;
      else if (A.src == 3 && A.in_bf) { kind = PSM_PAIR_UPCAT; c0 = u->convs[i - 1].cout; c1 = u->convs[A.skip].cout; }
      if (kind < 0 || !psm_pair_kernel_available(kind, cm, c0, c1, B.fuse_head)) continue;
      {   // the tile descriptors hold 32-bit byte offsets into the (zero-haloed) tensors of the whole case batch
        int64_t big = padded_bytes(H, W, cm, max_cases);
        if (kind == PSM_PAIR_POOL) big = std::max(big, padded_bytes(2 * H, 2 * W, c0, max_cases));
        if (kind == PSM_PAIR_UPCAT) big = std::max(big, std::max(padded_bytes(H / 2, W / 2, c0, max_cases), padded_bytes(H, W, c1, max_cases)));
        if (big >= ((int64_t)1 << 31)) continue;
      }
      if (cm == 32 && getenv("PSM_UNET_PAIR32") && atoi(getenv("PSM_UNET_PAIR32")) == 0) continue;     // diagnostic: 16-channel pairs only
      A.pair = 1; A.pair_kind = kind; B.pair = 2;
    }
    // in-workgroup K split (psm_conv3x3_kernel<..., KW = 2>): generic launches on finished bf16 inputs, 8-row tiles, two or more chunks, no split over workgroups
    {
      // Rule (profiles/r06_conv_experiments.txt (8)): four or more chunks AND at most one workgroup per CU -- an eight-wave workgroup holds 128 KB of LDS, so a
      // launch of more than 256 of them runs in two rounds (dec2a at 8 cases, 512 workgroups: 11.6 -> 14.3 us), and a two-chunk layer has nothing to pipeline
      // in a half (enc2b 6.0 -> 9.2 us).  PSM_UNET_KW=0 switches it off, PSM_UNET_KW=-n forces it for every eligible layer of n or more chunks (diagnostic).
      const int kw_env = getenv("PSM_UNET_KW") ? atoi(getenv("PSM_UNET_KW")) : 1;
      for (Conv& c : u->convs) {
        const bool can = u->bf16 && c.k == 3 && !c.stem && c.pair == 0 && !c.fuse_head && c.arrangement == 0 && c.in_bf && c.ksplit == 1 && c.n_chunks >= 2 && !c.x6;
        const int H = ny >> c.level, W = nx >> c.level;
        const long wgs = (long)((W + 15) / 16) * ((H + 7) / 8) * c.groups * max_cases;
        c.kw = !can || kw_env == 0 ? 1 : kw_env < 0 ? (c.n_chunks >= -kw_env ? 2 : 1) : (c.n_chunks >= 4 && wgs <= 256 ? 2 : 1);
      }
    }
  }
  // bf16 tensors (finished activations of bf16 mode; a fused pair always writes bf16) get the zero halo
  if (pass == 1) for (Conv& c : u->convs) act_layout(c, ny >> c.level, nx >> c.level, u->bf16 && c.k == 3 && (c.out_bf || c.pair != 0));
  for (Conv& c : u->convs) {
    const int H = ny >> c.level, W = nx >> c.level;
    const size_t ci = &c - u->convs.data();
    const bool feeds_conv3 = ci + 1 < u->convs.size() && u->convs[ci + 1].k == 3;
    if (pass == 1) {
      int rc = upload_conv(u, c);
      if (rc) return rc;
      if (c.pair == 1 && (rc = build_pair_tiles(u, c, u->convs[ci + 1], c.src != 0 ? &u->convs[ci - 1] : nullptr, c.src == 3 ? &u->convs[c.skip] : nullptr,
                                                H, W, max_cases))) return rc;
      free_dev(c.d_out); c.d_out = nullptr;
      c.slab = (int64_t)max_cases * H * W * c.cout;
      if (c.padded) {
        const size_t bytes = (size_t)max_cases * c.case_elems * sizeof(uint16_t) + 4096;      // + slack: a tile's last LDS-DMA piece may start past its last row
        UCHK(u, psm_dev_malloc((void**)&c.d_out, bytes));
        UCHK(u, hipMemsetAsync(c.d_out, 0, bytes, u->stream));
        UCHK(u, hipStreamSynchronize(u->stream));
      } else UCHK(u, psm_dev_malloc((void**)&c.d_out, (size_t)c.ksplit * c.slab * sizeof(float) + 64));
      continue;
    }
    const bool stem_layer = c.k == 3 && c.src == 0 && 9 * c.cin <= 64 && c.cout <= 16 && getenv("PSM_UNET_NO_STEM") == nullptr;
    // x6 pays where the matrix work dominates: measured at 8 cases per step it halves the wide layers (dec3a 66 -> 43 us,
    // enc4b 31 -> 20 us) and loses on the 16-channel 256^2 layers (enc0b 30 -> 67 us: half of every 32-channel chunk is
    // padding and the three-plane tiles leave one workgroup per CU) -- rule: c_in >= 64; psm_unet_autotune measures the rest
    const int x6c = ci < u->x6_choice.size() ? u->x6_choice[ci] : -1;
    const bool x6_ok = !u->bf16 && u->x6 && c.k == 3 && !stem_layer && c.src != 0 && (x6c == 1 || (x6c < 0 && c.cin >= 64));
    if (c.k == 3) choose_config(c, H, W, max_cases, feeds_conv3 && getenv("PSM_UNET_NO_SPLIT") == nullptr, u->bf16 ? 32 : 16,
                                ci < u->ksplit_cap.size() ? u->ksplit_cap[ci] : 8, x6_ok,
                                ci < u->tile_choice.size() ? u->tile_choice[ci] : -1,      // measured choice of psm_unet_autotune
                                u->bf16 && !stem_layer && c.src != 0,
                                (c.src != 0 && ci > 0 && u->convs[ci - 1].ksplit > 1) || (c.src == 3 && u->convs[c.skip].ksplit > 1));   // producers are planned before their consumers
    // diagnostic override: PSM_UNET_FORCE="layer:arrangement:nct:ksplit,..." (tools/attic/unet_bench.py experiments)
    if (const char* f = getenv("PSM_UNET_FORCE")) {
      for (const char* q = f; q && *q; q = std::strchr(q, ',') ? std::strchr(q, ',') + 1 : nullptr) {
        int li, arr, nct, ks;
        if (std::sscanf(q, "%d:%d:%d:%d", &li, &arr, &nct, &ks) == 4 && li == (int)ci && c.k == 3 && psm_conv_tile_rows(arr) &&
            nct == psm_conv_tile_nct(arr, nct) && nct <= (c.cout + 15) / 16 && ks >= 1 && ks <= 8 && (ks == 1 || feeds_conv3) && ks <= c.n_chunks) {
          c.arrangement = arr; c.nct = nct; c.groups = ((c.cout + 15) / 16 + nct - 1) / nct;
          c.x6 = x6_ok && arr == 0;
          c.n_chunks = (c.cin + (c.x6 || u->bf16 ? 32 : 16) - 1) / (c.x6 || u->bf16 ? 32 : 16);
          c.ksplit = std::min(ks, c.n_chunks);
        }
      }
    }
    c.stem = c.k == 3 && c.src == 0 && 9 * c.cin <= 64 && c.cout <= 16 && getenv("PSM_UNET_NO_STEM") == nullptr;
    if (c.stem) { c.ksplit = 1; c.nct = 1; c.groups = 1; c.arrangement = 0; c.x6 = false; }
    c.fuse_head = c.k == 3 && ci + 1 < u->convs.size() && u->convs[ci + 1].k == 1 && c.cout == 16 && !c.stem &&
                  getenv("PSM_UNET_NO_HEAD_FUSION") == nullptr;
    if (c.fuse_head) { c.arrangement = 0; c.nct = 1; c.groups = 1; c.ksplit = 1; c.x6 = x6_ok; c.n_chunks = (c.cin + (c.x6 || u->bf16 ? 32 : 16) - 1) / (c.x6 || u->bf16 ? 32 : 16); }
  }
  }
  free_dev(u->d_in); free_dev(u->d_field);
  if (u->h_in) { (void)hipHostFree(u->h_in); u->h_in = nullptr; }
  if (u->h_out) { (void)hipHostFree(u->h_out); u->h_out = nullptr; }
  const size_t npix = (size_t)ny * nx * max_cases;
  UCHK(u, psm_dev_malloc((void**)&u->d_in, npix * u->c_in * sizeof(float)));
  UCHK(u, psm_dev_malloc((void**)&u->d_field, npix * u->c_out * sizeof(float)));
  UCHK(u, hipHostMalloc((void**)&u->h_in, npix * u->c_in * sizeof(float), hipHostMallocDefault));
  UCHK(u, hipHostMalloc((void**)&u->h_out, npix * u->c_out * sizeof(float), hipHostMallocDefault));
  u->planned = true;
  return PSM_OK;
}

int psm_unet_forward_device(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, void* stream) {
  if (!u) return PSM_ERR_ARG;
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  if (!d_grid || !d_field) return fail(u, PSM_ERR_ARG, "null buffer");
  if (n_cases < 1 || n_cases > u->max_cases) return fail(u, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  UCHK(u, hipSetDevice(u->device));
  return forward(u, d_grid, n_cases, d_field, stream ? (hipStream_t)stream : u->stream);
}

int psm_unet_forward(psm_unet* u, const float* grid, int32_t n_cases, float* field) {
  if (!u) return PSM_ERR_ARG;
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  if (!grid || !field) return fail(u, PSM_ERR_ARG, "null buffer");
  if (n_cases < 1 || n_cases > u->max_cases) return fail(u, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  UCHK(u, hipSetDevice(u->device));
  const size_t npix = (size_t)u->ny * u->nx * n_cases;
  std::memcpy(u->h_in, grid, npix * u->c_in * sizeof(float));
  UCHK(u, hipMemcpyAsync(u->d_in, u->h_in, npix * u->c_in * sizeof(float), hipMemcpyHostToDevice, u->stream));
  int rc = forward(u, u->d_in, n_cases, u->d_field, u->stream);
  if (rc) return rc;
  UCHK(u, hipMemcpyAsync(u->h_out, u->d_field, npix * u->c_out * sizeof(float), hipMemcpyDeviceToHost, u->stream));
  {   // synchronous call of a few hundred us: poll instead of sleeping (see psm_api_solve.cpp, wait_stream); PSM_SYNC_BLOCK=1 blocks
    static const bool block = getenv("PSM_SYNC_BLOCK") != nullptr;
    hipError_t e = hipSuccess;
    if (block) e = hipStreamSynchronize(u->stream);
    else while ((e = hipStreamQuery(u->stream)) == hipErrorNotReady) { }
    UCHK(u, e);
  }
  std::memcpy(field, u->h_out, npix * u->c_out * sizeof(float));
  return PSM_OK;
}

int psm_unet_synchronize(psm_unet* u) {
  if (!u) return PSM_ERR_ARG;
  UCHK(u, hipSetDevice(u->device));
  UCHK(u, hipStreamSynchronize(u->stream));
  return PSM_OK;
}

int psm_unet_read_activation(psm_unet* u, int32_t idx, float* dst, int64_t dst_floats) {
  if (!u || !dst) return PSM_ERR_ARG;
  if (!u->planned || u->last_cases < 1) return fail(u, PSM_ERR_STATE, "no forward pass yet");
  if (idx < 0 || idx + 1 >= (int)u->convs.size()) return fail(u, PSM_ERR_ARG, "activation index out of range (the head's output is the field)");
  const Conv& c = u->convs[idx];
  if (!u->keep_act && (c.pair == 1 || (c.pair == 2 && c.fuse_head)))
    return fail(u, PSM_ERR_STATE, "this activation stays on chip (fused level pair): call psm_unet_keep_activations(u, 1) before psm_unet_plan");
  const int64_t n = (int64_t)u->last_cases * (u->ny >> c.level) * (u->nx >> c.level) * c.cout;
  if (dst_floats < n) return fail(u, PSM_ERR_ARG, "destination too small");
  UCHK(u, hipSetDevice(u->device));
  UCHK(u, hipStreamSynchronize(u->stream));
  if (c.padded) {                 // stored as bf16 in a zero-haloed tensor (act_layout): take the image rows, widen
    const int H = u->ny >> c.level, W = u->nx >> c.level;
    std::vector<uint16_t> hb((size_t)u->last_cases * c.case_elems);
    UCHK(u, psm_copy_d2h(hb.data(), c.d_out, hb.size() * sizeof(uint16_t)));
    for (int cs = 0; cs < u->last_cases; ++cs)
      for (int y = 0; y < H; ++y) {
        const uint16_t* row = hb.data() + (size_t)cs * c.case_elems + c.origin + (size_t)y * c.P * c.cout;
        float* d = dst + ((size_t)cs * H + y) * W * c.cout;
        for (int q = 0; q < W * c.cout; ++q) { const uint32_t w = (uint32_t)row[q] << 16; std::memcpy(&d[q], &w, 4); }
      }
    return PSM_OK;
  }
  UCHK(u, psm_copy_d2h(dst, c.d_out, n * sizeof(float)));
  if (c.ksplit > 1) {            // partial-sum slabs: finish like the consumer's loader (slab order, bias, ReLU)
    std::vector<float> tmp(n);
    for (int s = 1; s < c.ksplit; ++s) {
      UCHK(u, psm_copy_d2h(tmp.data(), c.d_out + (int64_t)s * c.slab, n * sizeof(float)));
      for (int64_t q = 0; q < n; ++q) dst[q] += tmp[q];
    }
    for (int64_t q = 0; q < n; ++q) { const float v = dst[q] + c.b[q % c.cout]; dst[q] = v > 0.f ? v : 0.f; }
  }
  return PSM_OK;
}

int psm_unet_profile(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, float* ms, int32_t* wgs) {
  if (!u || !ms) return PSM_ERR_ARG;
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  if (!d_grid || !d_field || n_cases < 1 || n_cases > u->max_cases) return fail(u, PSM_ERR_ARG, "bad arguments");
  UCHK(u, hipSetDevice(u->device));
  std::vector<hipEvent_t> ev(u->convs.size() + 1);
  for (auto& e : ev) UCHK(u, hipEventCreate(&e));
  int rc = forward(u, d_grid, n_cases, d_field, u->stream, ev.data());
  if (rc == PSM_OK) {
    hipError_t e = hipStreamSynchronize(u->stream);
    if (e != hipSuccess) rc = fail(u, PSM_ERR_HIP, hipGetErrorString(e));
  }
  for (size_t i = 0; rc == PSM_OK && i < u->convs.size(); ++i) {
    (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    if (wgs) {
      const Conv& c = u->convs[i];
      const int H = u->ny >> c.level, W = u->nx >> c.level;
      const int th = psm_conv_tile_rows(c.arrangement);
      wgs[i] = c.k == 3 ? ((W + 15) / 16) * ((H + th - 1) / th) * c.groups * c.ksplit * n_cases : 0;
      if (c.pair == 1) wgs[i] = ((W + PSM_PAIR_TX - 1) / PSM_PAIR_TX) * ((H + PSM_PAIR_TY - 1) / PSM_PAIR_TY) * n_cases;
      if (c.pair == 2) wgs[i] = 0;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

// every dispatch of `steps` forward passes with its own begin / end stamps: per convolution that owns a launch, the samples in ms
static int unet_collect_samples(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, int32_t steps,
                                std::vector<std::vector<float>>& samp, char* names) {
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  if (!d_grid || !d_field || n_cases < 1 || n_cases > u->max_cases) return fail(u, PSM_ERR_ARG, "bad arguments");
  UCHK(u, hipSetDevice(u->device));
  UCHK(u, hipStreamSynchronize(u->stream));
  const size_t nc = u->convs.size();
  samp.assign(nc, {});
  if (names) std::memset(names, 0, nc * 64);
  PsmLaunchProbe probe;
  auto drain = [&]() -> int {
    UCHK(u, hipStreamSynchronize(u->stream));
    for (auto& r : probe.recs) {
      float t = 0.f;
      if (r.tag >= 0 && r.tag < (int)nc && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
        samp[r.tag].push_back(t);
        if (names && !names[(size_t)r.tag * 64]) {
          std::string nm(r.name);
          while (!nm.empty() && (nm[0] == '(' || nm[0] == ' ')) nm.erase(0, 1);
          while (!nm.empty() && (nm.back() == ')' || nm.back() == ' ')) nm.pop_back();
          std::snprintf(names + (size_t)r.tag * 64, 64, "%s", nm.c_str());
        }
      }
      probe.pool.push_back(r.e0); probe.pool.push_back(r.e1);
    }
    probe.recs.clear();
    return PSM_OK;
  };
  int rc = PSM_OK;
  psm_launch_probe = &probe;
  for (int i = 0; i < steps && rc == PSM_OK; ++i) {
    rc = forward(u, d_grid, n_cases, d_field, u->stream);
    if (rc == PSM_OK && (i % 16) == 15) rc = drain();
  }
  psm_launch_probe = nullptr;
  if (rc == PSM_OK) rc = drain(); else (void)hipStreamSynchronize(u->stream);
  for (auto& r : probe.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : probe.pool) (void)hipEventDestroy(e);
  return rc;
}

int psm_unet_time_kernels(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, int32_t steps, double* us, int32_t* launches,
                          char* names) {
  if (!u || !us || steps < 1) return PSM_ERR_ARG;
  std::vector<std::vector<float>> samp;
  int rc = unet_collect_samples(u, d_grid, n_cases, d_field, steps, samp, names);
  if (rc) return rc;
  for (size_t i = 0; i < samp.size(); ++i) {
    double tot = 0.0;
    for (float t : samp[i]) tot += t;
    us[i] = samp[i].empty() ? 0.0 : tot / (double)samp[i].size() * 1e3;
    if (launches) launches[i] = (int32_t)(samp[i].size() / (size_t)steps);
  }
  return PSM_OK;
}

// the same pass, per launch the MEDIAN and the 10th / 90th percentile of its dispatch durations (microseconds)
int psm_unet_time_kernels_q(psm_unet* u, const float* d_grid, int32_t n_cases, float* d_field, int32_t steps, double* median_us,
                            double* p10_us, double* p90_us, int32_t* launches, char* names) {
  if (!u || !median_us || steps < 1) return PSM_ERR_ARG;
  std::vector<std::vector<float>> samp;
  int rc = unet_collect_samples(u, d_grid, n_cases, d_field, steps, samp, names);
  if (rc) return rc;
  for (size_t i = 0; i < samp.size(); ++i) {
    std::vector<float>& v = samp[i];
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    auto q = [&](double f) { return n ? (double)v[std::min(n - 1, (size_t)(f * (double)(n - 1) + 0.5))] * 1e3 : 0.0; };
    median_us[i] = q(0.5);
    if (p10_us) p10_us[i] = q(0.1);
    if (p90_us) p90_us[i] = q(0.9);
    if (launches) launches[i] = (int32_t)(n / (size_t)steps);
  }
  return PSM_OK;
}

// Plan-time autotune of the split-K depth: the planner splits the input channels of a layer that cannot fill the chip over up
// to 8 workgroups, which shortens that layer but makes its consumer sum float32 partial-sum slabs; whether the pair comes
// out ahead depends on both layers' shapes.  Measured, not guessed: for every split layer, in network order, halve the cap
// while the whole forward pass (median of `iters` passes of n_cases zero images) gets faster by more than 1 %.
int psm_unet_autotune(psm_unet* u, int32_t n_cases, int32_t iters, float* us_before, float* us_after) {
  if (!u) return PSM_ERR_ARG;
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  if (n_cases < 1 || n_cases > u->max_cases || iters < 1) return fail(u, PSM_ERR_ARG, "bad arguments");
  UCHK(u, hipSetDevice(u->device));
  const int ny = u->ny, nx = u->nx, mc = u->max_cases;
  auto measure = [&](double* out) -> int {
    UCHK(u, hipMemsetAsync(u->d_in, 0, (size_t)ny * nx * n_cases * u->c_in * sizeof(float), u->stream));
    hipEvent_t e0, e1;
    UCHK(u, hipEventCreate(&e0)); UCHK(u, hipEventCreate(&e1));
    int rc = PSM_OK;
    std::vector<float> t;
    for (int rep = 0; rep < 5 && rc == PSM_OK; ++rep) {           // 5 batches of `iters` passes, median batch
      for (int w = 0; w < 2 && rc == PSM_OK; ++w) rc = forward(u, u->d_in, n_cases, u->d_field, u->stream);
      (void)hipEventRecord(e0, u->stream);
      for (int i = 0; i < iters && rc == PSM_OK; ++i) rc = forward(u, u->d_in, n_cases, u->d_field, u->stream);
      (void)hipEventRecord(e1, u->stream);
      if (hipEventSynchronize(e1) != hipSuccess) rc = fail(u, PSM_ERR_HIP, "autotune: synchronize failed");
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      t.push_back(ms * 1e3f / iters);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    std::sort(t.begin(), t.end());
    *out = t[t.size() / 2];
    return PSM_OK;
  };
  const size_t nc = u->convs.size();
  if (u->ksplit_cap.size() != nc) u->ksplit_cap.assign(nc, 8);
  if (u->tile_choice.size() != nc) u->tile_choice.assign(nc, -1);
  if (u->pair_choice.size() != nc) u->pair_choice.assign(nc, -1);
  if (u->x6_choice.size() != nc) u->x6_choice.assign(nc, -1);
  double best = 0.0;
  int rc = measure(&best);
  if (rc) return rc;
  if (us_before) *us_before = (float)best;
  auto try_set = [&](std::vector<int>& vec, size_t i, int value) -> int {       // keep `value` if the whole pass gets > 1 % faster
    const int old = vec[i];
    vec[i] = value;
    int r = psm_unet_plan(u, ny, nx, mc);
    double t = 0.0;
    if (!r) r = measure(&t);
    if (r) return r;
    if (t < best * 0.99) { best = t; return PSM_OK; }
    vec[i] = old;
    return psm_unet_plan(u, ny, nx, mc);
  };
  // 1. fused level pairs (bf16 mode): the other choice than the tile-count rule made, per level that has a pair kernel
  for (size_t i = 0; u->bf16 && i + 1 < nc; ++i) {
    const Conv& A = u->convs[i]; const Conv& B = u->convs[i + 1];
    if (A.k != 3 || B.k != 3 || B.src != 1 || A.level != B.level || B.cout != A.cout || (A.cout != 16 && A.cout != 32)) continue;
    if ((rc = try_set(u->pair_choice, i, A.pair == 1 ? 0 : 1))) return rc;
  }
  // Steps 2 and 3 run twice: a shallower split of a producer changes what its consumer can run (finished bf16 inputs: the
  // in-workgroup K split, the 8-row x 16-channel tile of the long layers), which only a second look at the tiles can pick up.
  for (int sweep = 0; sweep < 2; ++sweep) {
  const double best_at_sweep = best;
  // 2. tile shape of the unfused 3x3 layers: the two candidates the planner did not pick
  for (size_t i = 0; i < nc; ++i) {
    const Conv& c = u->convs[i];
    if (c.k != 3 || c.stem || c.pair != 0 || c.fuse_head) continue;
    const int cur = c.arrangement == 2 ? 3 : c.arrangement == 1 ? 2 : (c.nct == 2 ? 0 : 1);
    for (int cand = 0; cand < 3; ++cand) {
      static const int NCT[3] = {2, 1, 4};
      if (cand == cur || NCT[cand] > (c.cout + 15) / 16) continue;
      if ((rc = try_set(u->tile_choice, i, cand))) return rc;
    }
  }
  // 2b. float32 mode: the other arithmetic (x6 on the bf16 matrix pipe / float32 MFMA) for every layer with an 8-row tile
  for (size_t i = 0; !u->bf16 && u->x6 && i < nc; ++i) {
    const Conv& c = u->convs[i];
    if (c.k != 3 || c.stem || c.src == 0 || c.arrangement != 0) continue;
    if ((rc = try_set(u->x6_choice, i, c.x6 ? 0 : 1))) return rc;
  }
  // 3. split-K depth
  for (size_t i = 0; i < u->convs.size(); ++i) {
    while (u->convs[i].ksplit > 1) {
      const int old_cap = u->ksplit_cap[i], old_split = u->convs[i].ksplit, cand = old_split / 2;
      u->ksplit_cap[i] = cand;
      if ((rc = psm_unet_plan(u, ny, nx, mc))) return rc;
      if (u->convs[i].ksplit >= old_split) {                      // the cap did not bind (a layer pinned by PSM_UNET_FORCE): the plan is the
        u->ksplit_cap[i] = old_cap;                               // one just measured -- re-measuring it would only chase timing noise
        if ((rc = psm_unet_plan(u, ny, nx, mc))) return rc;
        break;
      }
      double t = 0.0;
      if ((rc = measure(&t))) return rc;
      if (t < best * 0.99) { best = t; continue; }                // keep the shallower split, try one more halving
      u->ksplit_cap[i] = old_cap;                                  // no gain: back to what it was
      if ((rc = psm_unet_plan(u, ny, nx, mc))) return rc;
      break;
    }
  }
  if (best >= best_at_sweep) break;                                // nothing moved: a second sweep would only chase timing noise
  }
  if (us_after) *us_after = (float)best;
  return PSM_OK;
}

int psm_unet_get_choices(const psm_unet* u, int32_t* choices, int32_t n) {
  if (!u || !choices) return PSM_ERR_ARG;
  if (!u->planned) return PSM_ERR_STATE;
  const size_t nc = u->convs.size();
  if (n < (int32_t)(4 * nc)) return PSM_ERR_ARG;
  for (size_t i = 0; i < nc; ++i) {
    choices[4 * i + 0] = u->ksplit_cap.size() == nc ? u->ksplit_cap[i] : 8;
    choices[4 * i + 1] = u->tile_choice.size() == nc ? u->tile_choice[i] : -1;
    choices[4 * i + 2] = u->pair_choice.size() == nc ? u->pair_choice[i] : -1;
    choices[4 * i + 3] = u->x6_choice.size() == nc ? u->x6_choice[i] : -1;
  }
  return (int)nc;
}

int psm_unet_set_choices(psm_unet* u, const int32_t* choices, int32_t n) {
  if (!u || !choices) return PSM_ERR_ARG;
  if (!u->planned) return fail(u, PSM_ERR_STATE, "psm_unet_plan has not been called");
  const size_t nc = u->convs.size();
  if (n != (int32_t)(4 * nc)) return fail(u, PSM_ERR_ARG, "choices must hold 4 values per convolution");
  for (size_t i = 0; i < nc; ++i) {
    const int32_t* c = choices + 4 * i;
    if (c[0] < 1 || c[0] > 8 || c[1] < -1 || c[1] > 2 || c[2] < -1 || c[2] > 1 || c[3] < -1 || c[3] > 1)
      return fail(u, PSM_ERR_ARG, "choice out of range (split cap 1..8, tile -1..2, pair -1..1, x6 -1..1)");
  }
  u->ksplit_cap.assign(nc, 8); u->tile_choice.assign(nc, -1); u->pair_choice.assign(nc, -1); u->x6_choice.assign(nc, -1);
  for (size_t i = 0; i < nc; ++i) {
    u->ksplit_cap[i] = choices[4 * i]; u->tile_choice[i] = choices[4 * i + 1];
    u->pair_choice[i] = choices[4 * i + 2]; u->x6_choice[i] = choices[4 * i + 3];
  }
  UCHK(u, hipSetDevice(u->device));
  return psm_unet_plan(u, u->ny, u->nx, u->max_cases);
}

int psm_unet_ksplit(const psm_unet* u, int32_t idx) {
  if (!u || !u->planned || idx < 0 || idx >= (int)u->convs.size()) return PSM_ERR_ARG;
  return u->convs[idx].ksplit;
}

int psm_unet_plan_info(const psm_unet* u, int32_t idx, int32_t* info) {
  if (!u || !info || !u->planned || idx < 0 || idx >= (int)u->convs.size()) return PSM_ERR_ARG;
  const Conv& c = u->convs[idx];
  info[0] = psm_conv_tile_rows(c.arrangement); info[1] = c.nct; info[2] = c.ksplit; info[3] = c.pair | (c.x6 ? 4 : 0) | (c.kw == 2 ? 8 : 0);
  return PSM_OK;
}

int psm_unet_debug_run_layer(psm_unet* u, int32_t idx, float* stamps_us) {
  // diagnostic builds: re-run convolution `idx` alone on the activations of the last forward pass and return the
  // workgroup-0 stamps in microseconds after the first one (-1: not reached)
  if (!u || !stamps_us) return PSM_ERR_ARG;
  if (!u->planned || u->last_cases < 1 || idx < 0 || idx >= (int)u->convs.size()) return fail(u, PSM_ERR_STATE, "run a forward pass first");
  UCHK(u, hipSetDevice(u->device));
  std::vector<Conv> saved;
  // run the whole network but only stamp-read after it: stamps are overwritten by every layer, so run up to idx
  const size_t n_all = u->convs.size();
  (void)n_all;
  { unsigned long long z[64] = {0}; (void)z; }
  int rc = forward(u, u->d_in, u->last_cases, u->d_field, u->stream, nullptr, idx + 1);
  if (rc) return rc;
  UCHK(u, hipStreamSynchronize(u->stream));
  unsigned long long t[64];
  if (u->convs[idx].pair == 1) UCHK(u, psm_unet_pair_read_stamps(t));       // 3 workgroups x 16 stamps (first, middle, last)
  else UCHK(u, psm_unet_read_stamps(t));
  unsigned long long t0 = ~0ull;
  for (int k = 0; k < 64; ++k) if (t[k] && t[k] < t0) t0 = t[k];
  for (int k = 0; k < 64; ++k) stamps_us[k] = t[k] ? (float)((double)(t[k] - t0) * 0.01) : -1.f;
  return PSM_OK;
}

int64_t psm_unet_flops(const psm_unet* u) {
  if (!u || !u->planned) return 0;
  int64_t t = 0;
  for (const Conv& c : u->convs) t += 2LL * (u->ny >> c.level) * (u->nx >> c.level) * c.k * c.k * c.cin * c.cout;
  return t;
}

}  // extern "C"
