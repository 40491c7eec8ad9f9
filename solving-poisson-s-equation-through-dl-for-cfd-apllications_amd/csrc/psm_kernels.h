// psm_kernels.h -- launchers of the HIP kernels of one surrogate solve (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "psm_plan.h"

#include "psm_launch.h"

// Tiling constants shared by the host packers and the kernels -----------------
constexpr int PSM_PIX_PER_SLICE = 64;  // pixels of one block row handled by an encode workgroup
constexpr int PSM_MT_CHUNK = 4;        // 32-row M tiles staged in LDS at once
constexpr int PSM_STRIP_BAND = 16;     // block rows handled by one strips workgroup

struct PsmEncodeArgs {
  const float* grid;       // [cases, Ny, Nx, C_in]
  const float* mean;       // [K_in]
  const float4* bpack;     // [slices][NT][G][64] float4 (see pack_comp_in)
  float* part;             // [slices][Mpad][ldp]
  const int64_t* row_base; // [Mpad] float offset of each block's origin in grid, -1 = padding row
  int64_t row_stride;      // Nx*C_in floats
  int M, Mpad, NT, ldp, S, c_in, aligned;
  int whole;               // 33..128 rows: stage all rows at once (PSM_ENCODE_CHUNKED=1 keeps the double-buffered chunks)
  int x6;                  // float32 contraction as six bf16 MFMA terms of exactly split operands (psm_encode_x6_kernel)
  const uint4* bpack_x6;   // the basis pre-split into three bf16 planes, MFMA fragment order (pack_comp_in_x6); psm_encode_x6_mt_kernel only
  int kgroup;              // > 1: the M-tiled, wave-specialised form for large case batches (psm_encode_x6_mt_kernel) with `kgroup` K GROUPS: a
                           // workgroup owns one group of consecutive K slices x PSM_ENC_MT_ROWS (64) block rows and writes ONE slab -- part [kgroup][Mpad][ldp];
                           // 0 / 1: one slab per slice (psm_encode_x6_kernel / psm_encode_kernel)
  int pairs_ok;            // the two-slices-per-workgroup form may run (psm_encode_pair_kernel): when psm_encode_pairs(args) is true the launch writes n_slices / 2 slabs and the caller's reduce must sum that many (launch_all, psm_api_solve.cpp)
};
bool psm_encode_pairs(const PsmEncodeArgs& a);   // true: this launch writes n_slices / 2 slabs
constexpr int PSM_ENC_MT_ROWS = 64;    // block rows of a workgroup of the M-tiled encode (two 32-row MFMA tiles)

struct PsmReduceArgs {
  const float* part; float* xin;       // xin [Mpad][ldp]
  const float* ia; const float* ib;    // affine input scaler per column (0 on padding)
  int n_slices, Mpad, ldp;
};

struct PsmDenseArgs {
  const float* in; int ld_in;          // [Mpad][ld_in]
  const float* W; int ld_w;            // [Kpad][ld_w] zero padded (natural layout: fused reduce + layer-1 kernel)
  const void* Wp; int Kp;              // MFMA-packed copy [ld_w/16][Kp/16][64 lanes][4] (f32, or bf16 when bf16 != 0);
                                       // Kp = contraction length padded to 128 / 256 / a multiple of 512, zero rows beyond Kpad
  const float* bias;                   // [ld_w]
  const float* sa; const float* sb;    // head only: out = (acc+bias)*sa + sb
  float* out; int ld_out;              // [Mpad][ld_out]
  int Kpad, Mpad, relu, head;
  int bf16;                            // W points to bf16 [Kpad][ld_w]; activations rounded to bf16 on load
  int layer;                           // layer index (diagnostic stamps only)
  // LayerNormalization of the INPUT, fused into this launch (densePCA_attention, hidden layers): `in` holds the producer's raw
  // output v; every workgroup takes the moments of its own input rows (two passes over the first ln_n columns) and contracts with
  // (v - mean) * rsqrt(var + ln_eps) * ln_gamma + ln_beta.  ln_gamma / ln_beta: [>= ld_in], zero beyond ln_n; nullptr = plain input.
  // ln_residual: the epilogue adds the NORMALISED input at the output column (NNs.py:64 `x + attn_output`; square layer).
  const float* ln_gamma; const float* ln_beta; float ln_eps; int ln_n, ln_residual;
  // Large case batches (round 6): hidden activations in MFMA operand order instead of rows -- [row tile of 16][k group of 16][lane][4]
  // floats, lane = 16 * ((k % 16) / 4) + row % 16 -- so that the consumer's 16-byte-per-lane row loads are one contiguous KiB per wave
  // (a row-major [Mpad][512] activation gives sixteen 64-byte pieces of sixteen rows per load).  in_packed: `in` is such a tensor of
  // ld_in / 16 groups; out_packed: `out` is written as one of ld_out / 16 groups.  Plain float32 layers of 32-row tiles only.
  int in_packed, out_packed;
  // the strip-dot riders of the head launch read whole rows: the last hidden layer of a packed chain writes a second, row-major copy
  // (out_rows, [Mpad][ld_out]) beside the packed one, and the head launch hands it to its riders as in_rows ([Mpad][ld_in]); else nullptr
  float* out_rows; const float* in_rows;
};
// float offset of element (row, k) of a packed activation with `groups` k groups per row tile
static inline __host__ __device__ long long psm_packed_offset(int row, int k, int groups) {
  return ((((long long)(row >> 4) * groups + (k >> 4)) * 64) + ((k & 15) >> 2) * 16 + (row & 15)) * 4 + (k & 3);
}

// LayerNormalization of the reference's densePCA_attention (NNs.py:56, 64; Keras defaults: last axis, epsilon 1e-3, centre and
// scale), in place on a finished activation, with the optional residual of NNs.py:64 (`x + attn_output`, where attn_output
// is the INPUT of the Dense layer that produced x):  act[m][:] = LN(act[m][:] + res[m][:]) * gamma + beta over the first n
// columns.  Its own launch: a Dense launch tiles the output columns over workgroups (16 per workgroup, so that 32+ CUs pull
// the weights), no workgroup sees a whole row, and on this chip the kernel boundary is the cheapest grid-wide
// synchronisation; one wave per row, three passes over a row that sits in L2.
struct PsmLayerNormArgs {
  float* act; int ld_act;               // [rows][ld_act], normalised in place
  const float* res; int ld_res;         // [rows][ld_res] or null
  const float* gamma; const float* beta;   // [n]
  int rows, n;
  float eps;
};
hipError_t psm_launch_layernorm(const PsmLayerNormArgs& a, hipStream_t s);

// Conv1D layer of the reference's conv1D_PCA head (NNs.py:75-124): 'same' padding, cross-correlation like Keras,
// out[m][p][co] = act(bias[co] + sum_{t, ci} in[m][p + t - (k-1)/2][ci] * W[t][ci][co])  over the p_in scaled PCA coefficients
struct PsmConv1dArgs {
  const float* in; int64_t in_stride;   // row m at in + m * in_stride, element (p, ci) at p * c_in + ci
  const float* W; const float* bias;    // Keras Conv1D kernel [k][c_in][c_out], bias [c_out]
  float* out; int64_t out_stride;       // element (p, co) at p * c_out + co
  int M, P, k, c_in, c_out, relu;
};
hipError_t psm_launch_conv1d(const PsmConv1dArgs& a, hipStream_t s);

struct PsmDecodeArgs {
  const float* res; int ld_res;        // [Mpad][ld_res] inverse-scaled network output
  const float4* bpack;                 // [ncoltiles][Gd][64] float4
  const float* mean;                   // [K_out]
  const float* row_scale;              // [Mpad] out_scale per block row
  float* pred;                         // [M][K_out]
  int M, Mpad, Gd, n_coltiles, K_out;
  int x6;                              // batch decode: six bf16 MFMA terms of exactly split operands (float32 accuracy)
};

struct PsmStripArgs {
  const float* pred;                   // [cases*B][S*S*c_out]
  const float* grid;                   // [cases][Ny][Nx][c_in]
  const int32_t* strips;               // [B*NS (+S)][6]
  const int32_t* blk_y0x0;             // [B][2]
  float4* spart;                       // [cases][B][n_bands][NS] (sum f0, sum f1, count, -)
  float2* colpart;                     // gradp: [cases][n_bands][128] column sums of block 0 (else null)
  int NS, n_bands, B, S, c_in, c_out, sdf_ch, Ny, Nx;
};

struct PsmChainArgs {
  PsmChainParams cp;
  const PsmBlock* blocks;              // [B]
  const float4* spart; const float2* colpart; int n_bands;
  const float* pred;
  const int32_t* owner;                // [Ny*Nx]
  const int32_t* shiftA; const int32_t* shiftB;   // [c_out][Lmax] cell indices
  const int32_t* shiftOwnA; const int32_t* shiftOwnB;   // [c_out][Lmax] owner[cell]
  const float* shiftW;                 // [c_out][B] weight of each block's offset in the shift
  int shiftL[2]; int Lmax;
  float* offs;                         // [cases][c_out][B]
  float* shift;                        // [cases][c_out]
  int n_strips, c_out;
  unsigned long long* stamps;          // diagnostic build (-DPSM_STAMPS) only: s_memrealtime stamps of workgroup 0
};

struct PsmPasteArgs {
  const float* pred; const int32_t* owner; const float* offs; const float* shift;
  float* fields;                       // [cases][Ny][Nx][c_out]
  int B, S, c_out, npix;
};

// ---- geometry-bound fast path (psm_bind_geometry): every quantity the offset chain needs is linear in the network
// output, strip_sum(s) = scale * (act . g2[s] + c2[s]) with tables that depend on the geometry (flow-cell masks) and
// the model only -- so the strip means come from dot products with the LAST HIDDEN activation, computed by extra
// workgroups of the head layer's launch, and the decode launch runs the chain and pastes straight into the field:
// 6 launches per solve instead of 8, no decoded-block buffer.
struct PsmBindArgs {                   // table build, once per geometry
  const float* grid;                   // [Ny][Nx][c_in]: only the SDF channel is read
  const int32_t* strips;               // [nst][6]
  const int32_t* blk_y0x0;             // [B][2]
  const float* comp;                   // natural layout [ld_out][K_out] (rows >= p_out zero)
  const float* mean;                   // [K_out]
  const int32_t* owner;                // [Ny*Nx]
  const int32_t* shiftOwnA; const int32_t* shiftOwnB; int shiftL[2]; int Lmax;
  const float* Wh; int ldw; int Kh;    // head kernel, natural layout [Kh][ldw]
  const float* bh; const float* sa; const float* sb;
  double* G; double* Mrow;             // scratch [rows][ld_out], [rows]
  float* g2; float* c2; float* cnt;    // [rows][Kh], [rows], [rows]
  int32_t* row_of;                     // [rows] block whose activation row the dot uses
  uint32_t* ownbits;                   // [B][S*S/32] bit = this block-pixel is the last paste covering its cell
  int nst, B, S, c_in, c_out, sdf_ch, Ny, Nx, ld_out;
  int row_base;                        // case * B: row_of holds global block rows (case batches)
};
// Guard of the bound-geometry contract (psm.h: "the SDF channel of every solved grid must have the bound flow-cell
// pattern"): extra waves of the launch that computes the strip dots re-derive the pattern of the grid BEING SOLVED --
// one ballot per 64 consecutive pixels, 8 ballots per wave, 8 waves per guard workgroup -- and compare it with the bound one.
// A guard workgroup writes flags[workgroup] = 0 (match) or NaN (mismatch) on every solve (no reset needed; one flag per 4096
// pixels, so that the consumers sum 1024 flags for 64 cases, not 8192); the launch that writes the field adds the
// sum of the flags to the global shift, so a solve on another geometry returns NaN everywhere instead of a plausible
// wrong field, and *host_flag (mapped pinned memory) is raised for the host-side entries, which then fall back to the
// general path.  The riders run on CUs the head layer leaves idle: no launch, nothing on the critical path.  A large case
// batch re-reads 0.8 MB per case this way (64 cases: 1024 guard workgroups behind a 4 us layer cost it 6 us), so its guard
// workgroups are dealt over ALL Dense launches between the encode and the decode: [wg_first, wg_first + wg_count) per launch.
struct PsmGuardArgs {
  const float* sdf;                    // grid + sdf_channel (pixel stride c_in floats); null: no guard waves
  const unsigned long long* bits;      // [n_ballots] bound pattern, bit l of word g = pixel min(64 g + l, npix - 1) is a flow cell
  float* flags;                        // [ceil(n_waves / 8)]: one per guard workgroup
  int* host_flag;                      // device-side address of the workspace's word in mapped pinned memory (may be null)
  long long npix;                      // cases * Ny * Nx
  int c_in, n_ballots, n_waves;
  int wg_first, wg_count;              // the guard workgroups THIS launch carries
};
constexpr int PSM_GUARD_WG_WAVES = 8;  // guard waves per flag
constexpr int PSM_GUARD_BALLOTS = 8;   // per guard wave: 512 pixels
struct PsmDotsArgs {                   // rows: [c_out][nst] strip means, then [c_out][B] shift partial sums
  const float* g2; const float* c2; const float* cnt; const int32_t* row_of; const float* row_scale;
  float* out; int n_rows, Kh;
  PsmGuardArgs guard;
  // closed form of the chain (n_src > 1): a row is the concatenation of n_src table rows, one per source block of the
  // row's case -- out[r] = scale * (sum_blk act[case * n_src + blk] . g2[r * n_src + blk] + c2[r]); rows_per_case rows per case
  int n_src, rows_per_case;
};
struct PsmBoundArgs {
  PsmChainParams cp; const PsmBlock* blocks;
  const float* dots; const float* scnt; const uint32_t* ownbits; const int32_t* blk_y0x0; const float* shiftW;
  int shiftL[2];
  float* fields; float* offs; float* shift;
  int Nx, n_strips, B;
  const float* gflags; int n_gwaves;   // guard flags of this solve (0 / NaN), summed into the shift; never null (>= 1 entry)
  const float* cf_dots; const float* cf_a0; int cf;   // closed form of the chain (see PsmBoundBatchArgs): [C][B] each
  uint32_t field_bytes;                // size of `fields` in bytes when it is below 4 GiB (else 0): the paste then stores through a buffer
                                       // descriptor and switches a lane off by an out-of-range offset instead of a branch per value
};
struct PsmBoundBatchArgs {             // case batches: chain in its own small launch, then decode + paste
  PsmChainParams cp; const PsmBlock* blocks;
  const float* dots; const float* scnt;        // per case: [rows_pc] means / shift sums, [rows_pc] counts
  const uint32_t* ownbits;                     // [cases][B][S*S/32]
  const int32_t* blk_y0x0; const float* shiftW;
  int shiftL[2];
  float* fields; float* offs; float* shift;    // offs [cases][c_out][B], shift [cases][c_out]
  int Nx, npix, n_strips, B, rows_pc, n_cases;
  const float* gflags; int n_gwaves;           // as in PsmBoundArgs
  // Closed form of the offset chain (psm_bind_geometry_cases, B <= 64): on a bound geometry every branch of the chain is
  // decided by the strip counts, so offset + shift of block b is a fixed linear map of the strip means, which are
  // themselves dot products with the activation row of their source block: folded at bind time into one table row per
  // (field, block, source block), which the head launch contracts with the activation rows of the case's blocks in one
  // long dot per (field, block): cf_dots [cases][C][B].  The value subtracted from block b is cf_a0[case][f][b] + cf_dots
  // -- no chain launch between the head and this one, no chain waves in the single-case kernel.
  const float* cf_dots; const float* cf_a0; int cf;
  uint32_t field_bytes;                          // as in PsmBoundArgs
};
struct PsmPairFoldArgs {               // bind time: pair rows as linear combinations of the strip / shift rows
  const int32_t* ptr; const int32_t* src; const float* coef;    // CSR over the pair rows
  const float* g2; const float* c2;    // [rows][Kh], [rows] of one case
  float* g2p; float* c2p;              // [pairs][Kh], [pairs]
  int n_pairs, Kh;
};
hipError_t psm_launch_pair_fold(const PsmPairFoldArgs& a, hipStream_t s);
// strip dots from an arbitrary activation (introspection under the closed form; the bf16 handles' own dots launch)
hipError_t psm_launch_act_dots(const PsmDotsArgs& d, const float* act, int ld_act, int round_bf16, hipStream_t s, int packed = 0);
hipError_t psm_launch_bind(const PsmBindArgs& a, hipStream_t s);
hipError_t psm_launch_chain_dots(const PsmBoundBatchArgs& p, int c_out, hipStream_t s);
hipError_t psm_launch_decode_paste_batch(const PsmDecodeArgs& a, const PsmBoundBatchArgs& p, int c_out, hipStream_t s, int bf16 = 0);
// head layer + strip dots in one launch (f32, 16-row tiles)
hipError_t psm_launch_dense_dots(const PsmDenseArgs& a, const PsmDotsArgs& d, hipStream_t s);
// decode + offset chain + paste in one launch: ld_res <= 128, Mpad <= 64 (one case, B <= 64, n_x < 64)
hipError_t psm_launch_decode_paste(const PsmDecodeArgs& a, const PsmBoundArgs& p, int c_out, hipStream_t s, int bf16 = 0);
// bf16 handles: tables without the head layer folded in (Kh = ld_out), dots from the bf16-rounded `res`
hipError_t psm_launch_bind_unfolded(const PsmBindArgs& a, hipStream_t s);
hipError_t psm_launch_res_dots(const PsmDotsArgs& d, const float* res, int ld_res, hipStream_t s);

hipError_t psm_read_stamps(unsigned long long* out);   // [64]; zeros unless built with -DPSM_STAMPS
// ev_start / ev_stop (optional): stamped with the dispatch's own begin / end (hipExtLaunchKernel)
hipError_t psm_launch_encode(const PsmEncodeArgs& a, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// basis [slices][NT][KS/8][64] float4 (pack_comp_in) -> three bf16 planes in MFMA fragment order [slices][NT][KS/16][plane 3][64] uint4
hipError_t psm_launch_split_basis(const float4* bpack, uint4* out, int n_slices, int NT, int KS, hipStream_t s);
// bf16 operand path (psm_bf16.hip): bpack / weights point to bf16 data in the same tilings
hipError_t psm_launch_encode_bf16(const PsmEncodeArgs& a, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t psm_launch_decode_bf16(const PsmDecodeArgs& a, hipStream_t s);
hipError_t psm_launch_reduce(const PsmReduceArgs& a, hipStream_t s);
// slab reduce + first dense layer in one launch (ldp <= 512, layer width <= 1024)
hipError_t psm_launch_reduce_dense1(const PsmReduceArgs& r, const PsmDenseArgs& d, hipStream_t s);
hipError_t psm_launch_dense(const PsmDenseArgs& a, hipStream_t s, const PsmGuardArgs* riders = nullptr);   // riders: guard workgroups behind the layer's own
hipError_t psm_launch_decode(const PsmDecodeArgs& a, hipStream_t s);
hipError_t psm_launch_strips(const PsmStripArgs& a, int n_cases, hipStream_t s);
hipError_t psm_launch_chain(const PsmChainArgs& a, int n_cases, hipStream_t s);
hipError_t psm_launch_paste(const PsmPasteArgs& a, int n_cases, hipStream_t s);
// chain + shift + paste in one launch; valid when B <= 64 and n_x < 64
hipError_t psm_launch_assemble(const PsmChainArgs& a, const PsmPasteArgs& p, int n_cases, hipStream_t s);
// ring stage-in: grid from mapped pinned host memory -> device, + per-case out_scale (host, may be null) -> per-row scale
hipError_t psm_launch_stage_in(const float* src_host, float* dst, size_t n_floats, const float* scale_host, float* row_scale,
                               int n_rows, int B, hipStream_t s);
