// psm_unet.h -- launchers of the convolutional surrogate path (SURVEY.md §8 row a-conv; see psm_unet.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "psm_launch.h"

enum { PSM_SRC_SAME = 0, PSM_SRC_UPSAMPLE = 1, PSM_SRC_MAXPOOL = 2 };

struct PsmConvArgs {
  const float* in0;            // primary input, NHWC: [H][W][c0] (SAME), [H/2][W/2][c0] (UPSAMPLE), [2H][2W][c0] (MAXPOOL)
  const float* in1;            // optional skip input [H][W][c1], concatenated AFTER in0's channels (c1 = 0: none)
  const float4* wpack;         // [co_group][chunk][tap 9][ct NCT][lane 64] float4, see pack_conv3x3 (psm_unet_api.cpp)
  const float* bias;           // [cout_pad]
  float* out;                  // [H][W][cout]
  int c0, c1, n_chunks;        // channel chunks of 16 over the concatenated, zero-padded input
  // split-K producers: an input may arrive as `ks` partial-sum slabs (no bias, no ReLU yet); the loader adds
  // them in slab order, then the producer's bias and ReLU -- deterministic, and no reduction launch
  int ks0, ks1;                // slabs of in0 / in1 (1: a finished activation)
  int64_t slab0, slab1;        // slab strides (elements)
  const float* pbias0; const float* pbias1;   // producer biases (ks > 1), ReLU implied
  // fused 1x1 head (linear): when head_w != nullptr (cout == 16, one channel tile, finished output) the epilogue also
  // writes head_out[pixel][o] = sum_co act[co] * head_w[co][o] + head_b[o]
  const float* head_w; const float* head_b; float* head_out; int head_cout; int64_t head_case;
  int bf16;                    // operands rounded to bf16, chunks of 32 channels, wpack holds bf16x8 pieces
  int x6;                      // float32 mode on the bf16 matrix pipe: operands split exactly into three bf16 planes, six MFMA terms per product;
                               // chunks of 32 channels, wpack [co_group][chunk][plane 3][tap 9][ct][lane] bf16x8 (arrangement 0 only)
  // bf16 mode only: finished activations may live in HBM as bf16 (half the bytes of the bandwidth-bound shallow layers;
  // the consumer would round them to bf16 anyway, and rounding commutes with max-pool / upsample / concat).
  // in_bf: in0 and in1 are bf16 [..][c] (only with ks0 == ks1 == 1); out_bf: this layer stores bf16 (only with ksplit == 1)
  int in_bf, out_bf;
  int ksplit;                  // this layer's own split: workgroup z handles chunks [z*cps, (z+1)*cps) and writes slab z
  int kw;                      // 2: in-workgroup K split (eight waves, two halves of the workgroup's chunks, sums through LDS) -- bf16 activations, 8-row tiles, >= 2 chunks; else 1
  int64_t out_slab;
  int mode0;
  int H, W;                    // resolution of the convolution (its output)
  int H0, W0;                  // resolution of in0
  int cout, relu;
  int64_t in0_case, in1_case, out_case;   // per-case strides (elements)
  // row pitches in pixels (in0 at ITS resolution).  Finished bf16 activations live in zero-haloed tensors (psm_unet_api.cpp,
  // act_layout): the pointers above address pixel (0, 0) of case 0, a row is P pixels apart, a case *_case elements.  Dense
  // tensors (the raw image, float32 activations, split-K slabs): P = the tensor's width.
  int P0, P1, PO;
};

struct PsmHeadArgs {           // 1x1 convolution on a thin activation (c_in <= 64), linear
  const float* in; const float* w; const float* bias; float* out;   // w [c_in][c_out]
  int64_t n_pix; int c_in, c_out;
};

// arrangement (workgroup tile; 16 columns wide, 4 waves):
//   0 = 8 rows x NCT (1 or 2) channel tiles of 16, each wave 2 rows x all channel tiles
//   1 = 2 rows x 4 channel tiles, each wave both rows x one channel tile
// (round 4: 8 rows x 4 channel tiles with one channel tile per wave and the activation rows kept across ky -- 13 LDS operand
// reads per 24 MFMAs -- measured at 8 cases per step, bf16: slower on every deep layer, dec3a 21.7 against 15.3 us, with a split
// in two 14.3 us and its consumer +2.2 us; profiles/archive/r04_conv_experiments.txt; not kept)
// (4 x 16 and 8 x 16 pixel tiles with 4 channel tiles per wave were measured too: within 1 us per layer at batch 1,
// slower at 8 cases per step -- the staged bytes per MFMA are not what limits these layers; not kept)
// Round 6, bf16 activations only (finished bf16 inputs, no split-K slabs on the input side): larger per-wave register blocks, i.e. fewer
// LDS operand reads (ds_read_b128) per MFMA than the 1 : 1 of arrangement 0 / nct 2 --
//   2 = 16 rows x 4 channel tiles, each wave 4 rows x 4 channel tiles (64 px x 64 channels: 8 reads per 16 MFMAs)
//   3 = 16 rows x 2 channel tiles, each wave 4 rows x 2 channel tiles (6 reads per 8 MFMAs)
//   4 =  8 rows x 4 channel tiles, each wave 2 rows x 4 channel tiles (6 reads per 8 MFMAs; the input tile staged once for 64 channels)
inline int psm_conv_tile_rows(int arrangement) { return arrangement == 0 ? 8 : arrangement == 1 ? 2 : arrangement == 2 ? 16 : arrangement == 3 ? 16 : arrangement == 4 ? 8 : 0; }
inline int psm_conv_tile_nct(int arrangement, int nct) { return arrangement == 0 ? (nct == 2 ? 2 : 1) : arrangement == 3 ? 2 : 4; }
hipError_t psm_launch_conv3x3(const PsmConvArgs& a, int arrangement, int nct, int n_cases, hipStream_t st);
hipError_t psm_launch_head1x1(const PsmHeadArgs& a, hipStream_t st);
hipError_t psm_unet_read_stamps(unsigned long long* out);   // [64]; zeros unless built with -DPSM_STAMPS
// stem: first layer on the raw grid image, K = 9 * c_in flattened (c_in <= 7); wpack [ct][KG][lane] float4
hipError_t psm_launch_conv_stem(const PsmConvArgs& a, int n_cases, hipStream_t st);

// ---- fused level pairs (psm_unet_pair.hip): conv A + conv B of a level in one launch, bf16 mode ----------------------
#define PSM_PAIR_TX 30          // output tile of a workgroup (the mid tile is 32 x 16: two MFMA pixel tiles wide)
#define PSM_PAIR_TY 14
enum { PSM_PAIR_STEM = 0, PSM_PAIR_UPCAT = 1, PSM_PAIR_POOL = 2 };
// One fused-pair tile, host-built (psm_unet_api.cpp, build_pair_tiles): the kernel reads it with ONE scalar load per tile instead of
// two integer divisions and 64-bit address chains.  Offsets are BYTES from the tensors' (0, 0) pointers -- 32 bits: a case batch
// of activations is < 4 GB (checked at plan time).
struct PsmPairTile {
  int off0;     // in0: first staged source pixel of the tile (UPCAT: low-resolution pixel ((y0 - 2) / 2, (x0 - 2) / 2); POOL: (2 y0 - 4, 2 x0 - 4))
  int off1;     // in1: pixel (y0 - 2, x0 - 2)
  int offo;     // out / mid_out: pixel (y0, x0)
  int pix;      // dense pixel index cs * H * W + y0 * W + x0 (fused head's output, STEM's raw image)
  int y0, x0, cs;
  int flags;    // bit 0: every pixel the tile stages and computes lies inside the image
};
struct PsmPairArgs {
  const void* in0;             // STEM: float32 image [H][W][c0];  POOL: bf16 [2H][2W][c0];  UPCAT: bf16 [H/2][W/2][c0]
  const void* in1;             // UPCAT: bf16 skip [H][W][c1]
  const uint4* wA;             // conv A fragments (64 lanes x 16 bytes each), see pack_pair_a (psm_unet_api.cpp)
  const uint4* wB;             // conv B fragments
  const float* biasA; const float* biasB;
  unsigned short* out;         // bf16 [H][W][cm], or nullptr (head only)
  unsigned short* mid_out;     // introspection: conv A's activation, bf16 [H][W][cm], or nullptr
  const float* head_w; const float* head_b; float* head_out; int head_cout; int64_t head_case;   // fused linear 1x1 head (cm == 16)
  int H, W, c0, c1, cm;        // cm: channels of the level (conv A's and conv B's output)
  int tiles_x, tiles_y, n_cases;
  int64_t in0_case, in1_case, out_case;   // per-case strides (elements)
  int P0, P1, PO;              // row pitches in pixels of in0 (at its own resolution), in1, out / mid_out (see PsmConvArgs)
  const PsmPairTile* tiles;    // [n_cases * tiles_y * tiles_x], tile t = (cs * tiles_y + by) * tiles_x + bx
};
hipError_t psm_unet_pair_read_stamps(unsigned long long* out);   // [64]: 3 workgroups x 16 stamps; zeros unless built with -DPSM_STAMPS
bool psm_pair_kernel_available(int kind, int cm, int c0, int c1, bool head);   // shared by the planner and the launcher
hipError_t psm_launch_conv_pair(const PsmPairArgs& a, int kind, int cm, int n_cases, hipStream_t st);
