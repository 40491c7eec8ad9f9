// psm_mesh.hip -- mesh <-> uniform-grid ends of the per-step call:
// the solver hands over cells[N,5] = (Ux, Uy, Cx, Cy, p) in float64 (PythonComm.H:2-9) and
// reads back p[N] float64 (PythonComm.H:31-36).
//
//   umax     : U_max = max sqrt(Ux^2 + Uy^2)                               [PM:270]
//   to_grid  : barycentric mesh->grid interpolation, scatter into the image, normalisation,
//              SDF channel, NaN -> 0                                        [PM:272-297]
//   to_mesh  : gather at `indices`, barycentric grid->mesh interpolation with fill,
//              dimensionalise, near-wall / NaN fallback to the previous p   [PM:481-496]
//
// All of them are HBM-bound gathers; interpolation is done in float64 like the reference.
#include "psm_mesh.h"

#include <algorithm>

// U_max = max sqrt(Ux^2 + Uy^2) as NumPy takes it (PM:270), bit for bit, and bit for bit what the host pass of psm_solve takes
// (psm_api_mesh.cpp: max of the squares, one sqrt).  Found by tests/test_embed_host.py (round 6): the device reduction and the host
// pass gave U_max one ulp apart on 2 of 7 velocity scales -- and with it every pressure of those steps -- because hipcc contracts
// ux * ux + uy * uy into an fma in device code.  The squares and their sum are computed under `fp contract(off)`; the kernels
// reduce the SQUARED speed like the host does and take one square root at the end (sqrt_rn: the device's sqrt with a
// round-to-nearest correction from the exact residual -- a guard, no last-bit difference of sqrt itself was observed).
__device__ __forceinline__ double speed2_np(double ux, double uy) {
#pragma clang fp contract(off)               // (__dmul_rn / __dadd_rn are plain operators in this toolchain's headers and get contracted too)
  const double xx = ux * ux, yy = uy * uy;
  return xx + yy;
}
__device__ __forceinline__ double sqrt_rn(double x) {
  double r = sqrt(x);
  const unsigned long long eb = __double_as_longlong(r) & 0x7ff0000000000000ull;
  if (eb > (53ull << 52) && eb < 0x7ff0000000000000ull) {                       // normal, finite, not NaN
    const double u = __longlong_as_double(eb - (52ull << 52));                  // ulp(r)
    const double e = fma(-r, r, x);                                             // x - r^2, exact
    const double lim = r * u;                                                   // |sqrt(x) - r| <= u / 2  <=>  |e| <= r u (to 2nd order)
    if (e > lim) r += u; else if (e < -lim) r -= u;
  }
  return r;
}

__global__ __launch_bounds__(1024) void psm_umax_kernel(const double* cells, int64_t n, double* umax) {
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m = 0.0;
  for (int64_t i = tid; i < n; i += 1024) {
    const double ux = cells[i * 5], uy = cells[i * 5 + 1];
    const double v = speed2_np(ux, uy);
    m = (v > m || v != v) ? v : m;          // np.max propagates NaN
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_down(m, o, 64);
    m = (other > m || other != other) ? other : m;
  }
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double r = red[0];
    for (int w = 1; w < 16; ++w) r = (red[w] > r || red[w] != red[w]) ? red[w] : r;
    *umax = sqrt_rn(r);
  }
}

// np.max semantics on doubles: NaN propagates
__device__ __forceinline__ double nanmax2(double a, double b) { return (b > a || b != b) ? b : a; }

__global__ __launch_bounds__(1024) void psm_umax_partial_kernel(const double* cells, int64_t n, double* partials) {
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + tid; i < n; i += (int64_t)gridDim.x * 1024) {
    const double ux = cells[i * 5], uy = cells[i * 5 + 1];
    m = nanmax2(m, speed2_np(ux, uy));
  }
  for (int o = 32; o > 0; o >>= 1) m = nanmax2(m, __shfl_down(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double r = red[0];
    for (int w = 1; w < 16; ++w) r = nanmax2(r, red[w]);
    partials[blockIdx.x] = r;
  }
}

// First kernel of the one-graph psm_solve (registered caller buffers): reads the solver's cells[N,5] array STRAIGHT from host
// memory (the device-side address of the registered pages) -- every row requested at once over PCIe, no DMA-engine copy and no
// copy -> kernel dependency in front of the first kernel --, stores it to the device copy the gathers read, and takes the
// per-workgroup partial maxima of sqrt(Ux^2 + Uy^2) on the way (PM:270; reduced by every psm_to_grid workgroup).
typedef double psm_d2 __attribute__((ext_vector_type(2)));
// A workgroup owns 512 rows = 20480 bytes = 1280 16-byte pieces, five per thread, all five requested up front as fully
// coalesced 16-byte loads (a wave reads 1 KB of consecutive host memory per instruction: whole PCIe read requests, each line
// asked for once); the pieces go to the device copy from the registers and through LDS to the threads that own the rows'
// (Ux, Uy) for the maximum.  ALIGNED = false (a host array that is not 16-byte aligned): 8-byte pieces, same structure.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void psm_stage_cells_kernel(const double* host_cells, double* cells, int64_t n, double* partials) {
  constexpr int ROWS = 512, NP = ROWS * 5 / 2 / 256;        // 5 pieces of 16 bytes per thread
  __shared__ double row[ROWS * 5];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t dn = n * 5, n_chunks = (n + ROWS - 1) / ROWS;              // in doubles; chunks of 512 rows
  double m = 0.0;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {  // more than 256 chunks (131 072 cells): a second round
    const int64_t d0 = chunk * ROWS * 5;
    if (ALIGNED) {
      psm_d2 v[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int64_t e = d0 + 2 * (int64_t)(j * 256 + tid);
        v[j] = (e + 1 < dn) ? __builtin_nontemporal_load(reinterpret_cast<const psm_d2*>(host_cells + e))
                            : (psm_d2){e < dn ? __builtin_nontemporal_load(host_cells + e) : 0.0, 0.0};
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int64_t e = d0 + 2 * (int64_t)(j * 256 + tid);
        const int l = 2 * (j * 256 + tid);
        row[l] = v[j].x; row[l + 1] = v[j].y;
        if (e + 1 < dn) *reinterpret_cast<psm_d2*>(cells + e) = v[j];
        else if (e < dn) cells[e] = v[j].x;
      }
    } else {
      double v[2 * NP];
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j) {
        const int64_t e = d0 + (int64_t)(j * 256 + tid);
        v[j] = e < dn ? __builtin_nontemporal_load(host_cells + e) : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j) {
        const int64_t e = d0 + (int64_t)(j * 256 + tid);
        row[j * 256 + tid] = v[j];
        if (e < dn) cells[e] = v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ROWS / 256; ++k) {
      const int r = k * 256 + tid;
      if (chunk * ROWS + r < n) {
        const double ux = row[r * 5], uy = row[r * 5 + 1];
        m = nanmax2(m, speed2_np(ux, uy));
      }
    }
    __syncthreads();                                         // the rows are consumed: the next round may overwrite them
  }
  for (int o = 32; o > 0; o >>= 1) m = nanmax2(m, __shfl_down(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) partials[blockIdx.x] = nanmax2(nanmax2(red[0], red[1]), nanmax2(red[2], red[3]));
}

hipError_t psm_launch_stage_cells(const double* host_cells, double* cells, int64_t n, double* partials, int* n_partials, hipStream_t st) {
  if (n < 1) return hipErrorInvalidValue;
  const int64_t wgs = std::min<int64_t>(256, (n + 511) / 512);   // the partials array holds 256: larger meshes take further rounds
  *n_partials = (int)wgs;
  const bool aligned = ((reinterpret_cast<uintptr_t>(host_cells) | reinterpret_cast<uintptr_t>(cells)) & 15) == 0;
  if (aligned) hipLaunchKernelGGL(psm_stage_cells_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, st, host_cells, cells, n, partials);
  else hipLaunchKernelGGL(psm_stage_cells_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, st, host_cells, cells, n, partials);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void psm_to_grid_kernel(PsmToGridArgs a) {
  __shared__ double um_s[4];
  double umax_v = a.umax ? *a.umax : a.umax_val;
  if (a.umax_partials) {                      // uniform: every workgroup reduces the <= 256 partial maxima itself
    double m = a.umax_partials[min((int)threadIdx.x, a.n_partials - 1)];
    for (int o = 32; o > 0; o >>= 1) m = nanmax2(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) um_s[threadIdx.x >> 6] = m;
    __syncthreads();
    umax_v = sqrt_rn(nanmax2(nanmax2(um_s[0], um_s[1]), nanmax2(um_s[2], um_s[3])));      // the partials are maxima of the SQUARED speed
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.umax_out = umax_v;
  }
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= a.n_grid) return;
  const int src = a.src_of_cell[cell];     // grid point whose value lands in this cell (-1: never written -> 0)
  float ux = 0.f, uy = 0.f;
  if (src >= 0) {
    const double inv = 1.0 / umax_v;
    const int32_t* v = a.vtx + (int64_t)src * 3;
    const double* w = a.wts + (int64_t)src * 3;
    double sx = 0.0, sy = 0.0;
    bool neg = false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double* c = a.cells + (int64_t)v[j] * 5;
      sx += (c[0] * inv) * w[j];           // interpolate(Ux / U_max)  (PM:272,280)
      sy += (c[1] * inv) * w[j];
      neg = neg || (w[j] < 0.0);
    }
    if (a.fill && neg) { sx = NAN; sy = NAN; }   // interpolate_fill (SMD:421-423): NaN, then NaN -> 0
    sx /= a.max_abs_ux;                    // PM:290-291
    sy /= a.max_abs_uy;
    ux = (sx != sx) ? 0.f : (float)sx;     // grid[np.isnan(grid)] = 0  (PM:297)
    uy = (sy != sy) ? 0.f : (float)sy;
  }
  const double sd = a.sdf[cell] * a.sdf_scale;   // PM:292 (raw) / SMD:443 (divided by max_abs_dist)
  float* g = a.grid + cell * a.c_in;
  g[0] = ux;
  g[1] = uy;
  g[2] = (sd != sd) ? 0.f : (float)sd;
}

__global__ __launch_bounds__(256) void psm_to_mesh_kernel(PsmToMeshArgs a) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= a.n_cells) return;
  const int32_t* v = a.vtx + n * 3;
  const double* w = a.wts + n * 3;
  double acc = 0.0;
  bool neg = false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int cell = a.cell_of_point[v[j]];                         // p_adim_unif = result[indices]  (PM:481)
    acc += (double)a.field[(int64_t)cell * a.c_out] * w[j];
    neg = neg || (w[j] < 0.0);
  }
  const double um = a.umax ? *a.umax : a.umax_val;
  double p = acc * a.max_abs_p * (um * um);                         // PM:490
  const double prev = a.cells[n * 5 + 4];
  if (a.near_wall[n] || neg || acc != acc) p = prev;                // PM:494, 496 (interpolate_fill -> NaN)
  a.p_out[n] = p;
}

hipError_t psm_launch_umax_partial(const double* cells, int64_t n, double* partials, int* n_partials, hipStream_t st) {
  const int nwg = (int)std::min<int64_t>(256, (n + 4095) / 4096);
  *n_partials = nwg;
  hipLaunchKernelGGL(psm_umax_partial_kernel, dim3(nwg), dim3(1024), 0, st, cells, n, partials);
  return hipGetLastError();
}

hipError_t psm_launch_umax(const double* cells, int64_t n, double* umax, hipStream_t st) {
  hipLaunchKernelGGL(psm_umax_kernel, dim3(1), dim3(1024), 0, st, cells, n, umax);
  return hipGetLastError();
}
hipError_t psm_launch_to_grid(const PsmToGridArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_to_grid_kernel, dim3((unsigned)((a.n_grid + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}
// k columns of mesh values -> grid image, float64: out[cell][c] = interpolate(_fill)(values[:, c]) of the
// grid point that NumPy's fancy assignment leaves in that cell (last writer), 0 for cells never written
// (np.zeros base image); NaNs of interpolate_fill are kept (the caller's `grid[np.isnan(grid)] = 0`).
__global__ __launch_bounds__(256) void psm_interp_to_grid_kernel(const double* values, int k, const int32_t* vtx, const double* wts,
                                                                 const int32_t* src_of_cell, int fill, double* out, int64_t n_grid) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= n_grid) return;
  const int src = src_of_cell[cell];
  if (src < 0) {
    for (int c = 0; c < k; ++c) out[cell * k + c] = 0.0;
    return;
  }
  const int32_t* v = vtx + (int64_t)src * 3;
  const double* w = wts + (int64_t)src * 3;
  const bool neg = (w[0] < 0.0) || (w[1] < 0.0) || (w[2] < 0.0);
  for (int c = 0; c < k; ++c) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) s += values[(int64_t)v[j] * k + c] * w[j];      // np.einsum('nj,nj->n', take(values, vtx), wts)
    out[cell * k + c] = (fill && neg) ? NAN : s;
  }
}

hipError_t psm_launch_interp_to_grid(const double* values, int k, const int32_t* vtx, const double* wts, const int32_t* src_of_cell,
                                     int fill, double* out, int64_t n_grid, hipStream_t st) {
  hipLaunchKernelGGL(psm_interp_to_grid_kernel, dim3((unsigned)((n_grid + 255) / 256)), dim3(256), 0, st, values, k, vtx, wts,
                     src_of_cell, fill, out, n_grid);
  return hipGetLastError();
}

hipError_t psm_launch_to_mesh(const PsmToMeshArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_to_mesh_kernel, dim3((unsigned)((a.n_cells + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// separable Gaussian smoothing of an assembled field (SMD:353-363, UGP:366-367):
// scipy.ndimage.gaussian_filter(order=0, mode='reflect', truncate=4.0) = correlate1d along
// axis 0, then along axis 1, with weights exp(-x^2 / 2 sigma^2) / sum over |x| <= int(4 sigma + 0.5)
// and half-sample-symmetric ("reflect": d c b a | a b c d | d c b a) boundaries.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psm_gauss1d_kernel(const float* in, float* out, int ny, int nx, int axis,
                                                          int radius, const float* wts) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ny * nx) return;
  const int y = idx / nx, x = idx - y * nx;
  const int n = axis == 0 ? ny : nx, i = axis == 0 ? y : x, stride = axis == 0 ? nx : 1;
  const float* line = in + (axis == 0 ? x : y * nx);
  float acc = 0.f;
  for (int t = -radius; t <= radius; ++t) {
    int j = (i + t) % (2 * n);
    if (j < 0) j += 2 * n;
    if (j >= n) j = 2 * n - 1 - j;
    acc += wts[t + radius] * line[(int64_t)j * stride];
  }
  out[idx] = acc;
}

hipError_t psm_launch_gauss1d(const float* in, float* out, int ny, int nx, int axis, int radius, const float* wts, hipStream_t st) {
  hipLaunchKernelGGL(psm_gauss1d_kernel, dim3((ny * nx + 255) / 256), dim3(256), 0, st, in, out, ny, nx, axis, radius, wts);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// a8 (evaluation only): label blocks with the per-block mean over the flow cells removed --
//   y_array[step, ..., c][x_array[step, ..., sdf] != 0] -= mean(y_array[step, ..., c][x_array[step, ..., sdf] != 0])
// (SM_call.py:487-488; Eval_dual_Dense_onlycil.py:509-511).  One workgroup per (block, channel); float64 sums like the
// float64 grid of the reference.  A block without flow cells keeps its values (the reference's empty-slice mean is
// NaN but is assigned to an empty selection).
__global__ __launch_bounds__(256) void psm_label_blocks_kernel(const float* grid, const float* labels, const int32_t* blk_y0x0,
                                                               float* out, int S, int c_in, int c_out, int sdf_ch, int Nx) {
  const int b = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  const int y0 = blk_y0x0[2 * b], x0 = blk_y0x0[2 * b + 1];
  __shared__ double ssum[256];
  __shared__ double scnt[256];
  double sum = 0.0, cnt = 0.0;
  for (int i = t; i < S * S; i += 256) {
    const int64_t pix = (int64_t)(y0 + i / S) * Nx + x0 + i % S;
    if (grid[pix * c_in + sdf_ch] != 0.f) { sum += (double)labels[pix * c_out + c]; cnt += 1.0; }
  }
  ssum[t] = sum; scnt[t] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) { ssum[t] += ssum[t + s]; scnt[t] += scnt[t + s]; }
    __syncthreads();
  }
  const double mean = scnt[0] > 0.0 ? ssum[0] / scnt[0] : 0.0;
  for (int i = t; i < S * S; i += 256) {
    const int64_t pix = (int64_t)(y0 + i / S) * Nx + x0 + i % S;
    const double v = (double)labels[pix * c_out + c];
    out[((int64_t)b * S * S + i) * c_out + c] = (float)(grid[pix * c_in + sdf_ch] != 0.f ? v - mean : v);
  }
}

hipError_t psm_launch_label_blocks(const float* grid, const float* labels, const int32_t* blk_y0x0, float* out, int B, int S,
                                   int c_in, int c_out, int sdf_ch, int Nx, hipStream_t st) {
  hipLaunchKernelGGL(psm_label_blocks_kernel, dim3(B, c_out), dim3(256), 0, st, grid, labels, blk_y0x0, out, S, c_in, c_out, sdf_ch, Nx);
  return hipGetLastError();
}

// compute_in_block_error (pressureSM_deltas/utils.py:210-243; called at SM_call.py:555-557 on the decoded blocks BEFORE the
// reassembly): partial sums per workgroup over the flow cells of one block -- count and sum / sum of squares of the
// non-NaN differences pred - true, extrema of true and pred, count of NaN truths (np.max then gives NaN) -- float64 like
// the reference's arrays; `true` = label block * row_scale[b] (SM_call.py:555: y_array * max_abs_p * U_max_norm^2, the scale
// the decoded blocks already carry).  Partials [B][8] doubles, summed on the host.
__global__ __launch_bounds__(256) void psm_block_error_kernel(const float* grid, const float* pred, const float* label_blocks,
                                                              const float* row_scale, const int32_t* blk_y0x0, double* part,
                                                              int S, int c_in, int c_out, int sdf_ch, int Nx) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int y0 = blk_y0x0[2 * b], x0 = blk_y0x0[2 * b + 1];
  const double sc = (double)row_scale[b];
  double n = 0.0, s1 = 0.0, s2 = 0.0, tmin = INFINITY, tmax = -INFINITY, pmin = INFINITY, pmax = -INFINITY, tnan = 0.0;
  for (int i = t; i < S * S; i += 256) {
    const int64_t pix = (int64_t)(y0 + i / S) * Nx + x0 + i % S;
    if (!(grid[pix * c_in + sdf_ch] != 0.f)) continue;
    for (int c = 0; c < c_out; ++c) {
      const int64_t e = ((int64_t)b * S * S + i) * c_out + c;
      const double tr = (double)label_blocks[e] * sc, pr = (double)pred[e];
      if (tr != tr) tnan += 1.0; else { tmin = fmin(tmin, tr); tmax = fmax(tmax, tr); }
      if (pr == pr) { pmin = fmin(pmin, pr); pmax = fmax(pmax, pr); }
      const double d = pr - tr;
      if (d == d) { n += 1.0; s1 += d; s2 += d * d; }
    }
  }
  __shared__ double sh[8][256];
  sh[0][t] = n; sh[1][t] = s1; sh[2][t] = s2; sh[3][t] = tmin; sh[4][t] = tmax; sh[5][t] = pmin; sh[6][t] = pmax; sh[7][t] = tnan;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      sh[0][t] += sh[0][t + s]; sh[1][t] += sh[1][t + s]; sh[2][t] += sh[2][t + s]; sh[7][t] += sh[7][t + s];
      sh[3][t] = fmin(sh[3][t], sh[3][t + s]); sh[4][t] = fmax(sh[4][t], sh[4][t + s]);
      sh[5][t] = fmin(sh[5][t], sh[5][t + s]); sh[6][t] = fmax(sh[6][t], sh[6][t + s]);
    }
    __syncthreads();
  }
  if (t < 8) part[(int64_t)b * 8 + t] = sh[t][0];
}

hipError_t psm_launch_block_error(const float* grid, const float* pred, const float* label_blocks, const float* row_scale,
                                  const int32_t* blk_y0x0, double* part, int B, int S, int c_in, int c_out, int sdf_ch, int Nx, hipStream_t st) {
  hipLaunchKernelGGL(psm_block_error_kernel, dim3(B), dim3(256), 0, st, grid, pred, label_blocks, row_scale, blk_y0x0, part, S, c_in, c_out, sdf_ch, Nx);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// U_to_gradP: integration of (dp/dx, dp/dy) into p over four quadrants
// (integrate_field UGP:371-416, stitching UGP:597-628).  For a quadrant with reference corner
// column `ij` and row `ii` the reference's double loop reduces to
//     Phat[a,b] = (SdPy[a,ij] - SdPy[ii,ij]) + (SdPx[a,b] - SdPx[a,ij])
// i.e. one row-wise cumulative sum per row (with the reference's "reset at the obstacle" index
// quirk, precomputed per row on the host as (v, u) pairs: aaa[v] = -(ccc[v] - ccc[u])) and ONE
// column-wise cumulative sum per side.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void psm_block_scan256(float* sh, float& v, int tid) {   // inclusive scan of 256 values
  sh[tid] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const float t = tid >= o ? sh[tid - o] : 0.f;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  v = sh[tid];
}

__global__ __launch_bounds__(256) void psm_integ_rows_kernel(PsmIntegArgs a) {
  __shared__ float sh[256];
  __shared__ float fix[2 * PSM_INTEG_MAX_FIX + 2];
  const int tid = threadIdx.x, y = blockIdx.x;
  const int la = y < a.cy ? y : y - a.cy;                       // block-local row: indexes the fix-up table
  const int2* fx = a.fixups + (int64_t)la * PSM_INTEG_MAX_FIX;
  for (int side = 0; side < 2; ++side) {                        // 0: left block [0,cx), 1: right block [cx-1,nx)
    const int x0 = side == 0 ? 0 : a.cx - 1, w = side == 0 ? a.cx : a.nx - a.cx + 1;
    float* out = side == 0 ? a.rxl + (int64_t)y * a.cx : a.rxr + (int64_t)y * (a.nx - a.cx + 1);
    const float* g = a.gradp + ((int64_t)y * a.nx + x0) * 2;    // channel 0
    const int per = (w + 255) / 256, j0 = tid * per, j1 = min(w, j0 + per);
    float s = 0.f;
    for (int j = j0; j < j1; ++j) s += g[2 * j];
    float incl = s;
    psm_block_scan256(sh, incl, tid);
    float run = incl - s;                                       // exclusive prefix of this thread's chunk
    for (int j = j0; j < j1; ++j) { run += g[2 * j]; out[j] = run; }   // ccc
    __syncthreads();
    // "aaa[nn] = -dd": per distinct index v, delta_v = -(ccc[v] - ccc[u]) - aaa[v]
    if (tid < PSM_INTEG_MAX_FIX) {
      const int v = fx[tid].x, u = fx[tid].y;
      float d = 0.f;
      if (v >= 0 && v < w) d = -(out[v] - (u >= 0 ? out[u] : 0.f)) - g[2 * v];
      fix[tid] = d;
    }
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
      float add = 0.f;
#pragma unroll
      for (int e = 0; e < PSM_INTEG_MAX_FIX; ++e) add += (fx[e].x >= 0 && fx[e].x <= j) ? fix[e] : 0.f;
      out[j] += add;
    }
    __syncthreads();
    const float ref = out[side == 0 ? 0 : w - 1];               // SdPx[a, ij]
    __syncthreads();
    for (int j = j0; j < j1; ++j) out[j] = (out[j] - ref) * a.dx;
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void psm_integ_cols_kernel(PsmIntegArgs a) {
  // column cumulative sums of dp/dy at global columns 0 (left blocks) and nx-1 (right blocks),
  // restarted at the cut row; then the shift of the left quadrants onto the right ones
  extern __shared__ float sm[];                                  // [2][ny]
  const int tid = threadIdx.x;
  float* yl = sm; float* yr = sm + a.ny;
  if (tid < 4) {                                                 // 4 short serial scans (ny <= a few thousand)
    const int side = tid & 1, bottom = tid >> 1;
    const int r0 = bottom ? a.cy : 0, r1 = bottom ? a.ny : a.cy;
    const int col = side == 0 ? 0 : a.nx - 1;
    float* dst = side == 0 ? yl : yr;
    float run = 0.f;
    for (int y = r0; y < r1; ++y) { run += a.gradp[((int64_t)y * a.nx + col) * 2 + 1]; dst[y] = run; }
    const float ref = bottom ? dst[r1 - 1] : dst[r0];            // SdPy[ii, ij]
    for (int y = r0; y < r1; ++y) dst[y] = (dst[y] - ref) * a.dy;
  }
  __syncthreads();
  for (int y = tid; y < a.ny; y += 1024) { a.yl[y] = yl[y]; a.yr[y] = yr[y]; }
  __shared__ float red[2][16];
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = a.nx - a.cx + 1;
  for (int q = 0; q < 2; ++q) {                                  // top pair (blocks 2 vs 1), bottom pair (4 vs 3)
    const int n = a.npair[q];
    const int2* pr = a.pairs + (q == 0 ? 0 : a.npair[0]);
    float acc = 0.f;
    for (int k = tid; k < n; k += 1024) {
      const int rl = pr[k].x, rr = pr[k].y;                      // global rows
      acc += (yl[rl] + a.rxl[(int64_t)rl * a.cx + a.cx - 1]) - (yr[rr] + a.rxr[(int64_t)rr * wr]);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) red[q][wave] = acc;
  }
  __syncthreads();
  if (tid < 2) {
    float t = 0.f;
    for (int w8 = 0; w8 < 16; ++w8) t += red[tid][w8];
    a.corr[tid] = t / (float)a.npair[tid];                       // mean of an empty selection -> NaN like NumPy
  }
}

__global__ __launch_bounds__(256) void psm_integ_write_kernel(PsmIntegArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.ny * a.nx) return;
  const int y = idx / a.nx, x = idx - y * a.nx;
  float v;
  if (x < a.cx) v = a.yl[y] + a.rxl[(int64_t)y * a.cx + x] - a.corr[y < a.cy ? 0 : 1];
  else v = a.yr[y] + a.rxr[(int64_t)y * (a.nx - a.cx + 1) + (x - (a.cx - 1))];
  a.p_out[idx] = v;
}

hipError_t psm_launch_integrate(const PsmIntegArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_integ_rows_kernel, dim3(a.ny), dim3(256), 0, st, a);
  hipLaunchKernelGGL(psm_integ_cols_kernel, dim3(1), dim3(1024), (size_t)2 * a.ny * sizeof(float), st, a);
  hipLaunchKernelGGL(psm_integ_write_kernel, dim3((a.ny * a.nx + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}
