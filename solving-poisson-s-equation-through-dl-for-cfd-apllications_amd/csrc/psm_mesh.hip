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

__global__ __launch_bounds__(1024) void psm_umax_kernel(const double* cells, int64_t n, double* umax) {
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double m = 0.0;
  for (int64_t i = tid; i < n; i += 1024) {
    const double ux = cells[i * 5], uy = cells[i * 5 + 1];
    const double v = sqrt(ux * ux + uy * uy);
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
    *umax = r;
  }
}

__global__ __launch_bounds__(256) void psm_to_grid_kernel(PsmToGridArgs a) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= a.n_grid) return;
  const int src = a.src_of_cell[cell];     // grid point whose value lands in this cell (-1: never written -> 0)
  float ux = 0.f, uy = 0.f;
  if (src >= 0) {
    const double inv = 1.0 / *a.umax;
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
  const double um = *a.umax;
  double p = acc * a.max_abs_p * (um * um);                         // PM:490
  const double prev = a.cells[n * 5 + 4];
  if (a.near_wall[n] || neg || acc != acc) p = prev;                // PM:494, 496 (interpolate_fill -> NaN)
  a.p_out[n] = p;
}

hipError_t psm_launch_umax(const double* cells, int64_t n, double* umax, hipStream_t st) {
  hipLaunchKernelGGL(psm_umax_kernel, dim3(1), dim3(1024), 0, st, cells, n, umax);
  return hipGetLastError();
}
hipError_t psm_launch_to_grid(const PsmToGridArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(psm_to_grid_kernel, dim3((unsigned)((a.n_grid + 255) / 256)), dim3(256), 0, st, a);
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
