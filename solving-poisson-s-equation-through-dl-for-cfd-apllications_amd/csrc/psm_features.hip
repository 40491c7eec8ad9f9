// psm_features.hip -- input features of the pressureSM_Poisson surrogate
// (Improved_SM/deltaU_to_deltaP/source/pressureSM_Poisson/SM_call.py = SMP, lines 588-711):
//   channel 0: arcsinh-smoothed Poisson source term  (dUx/dx)^2 + 2 dUx/dy dUy/dx + (dUy/dy)^2, L^2/U^2
//              np.gradient differences (unit spacing, one-sided on the border), zero wherever the
//              cell or a direct neighbour is solid (SMP:602-632), then `smart_arcsin_smooth_transform`
//              (SMP:22-69): central range mean +- k std of the WHOLE image -> [-1,1], linear tails, arcsinh
//   channels 1, 2: delta U / U;  channel 3: signed-distance image;  each divided by its max_abs.
// Everything is float64 like the reference (differences of nearly equal velocities); the grid image is
// written as float32 NHWC, the layout the encode kernel reads.  Two launches: the term plus
// per-workgroup (sum, sum of squares) partials in a fixed order, then every workgroup of the
// second launch folds the partials the same way (deterministic) and transforms its pixels.
#include "psm_mesh.h"

namespace {
constexpr int FT = 256;

// d/dy and d/dx of one field at (y, x) like SMP:602-632: np.gradient (central inside, first-order
// one-sided on the border), both zero when the cell or a direct neighbour is NaN -- a cell is NaN
// when it is solid (sdfunct == 0, SMP:624-625) or already NaN in the interpolated field.
__device__ __forceinline__ void masked_grad(const double* __restrict__ u, const double* __restrict__ sdf, int y, int x,
                                            int ny, int nx, double& d_dy, double& d_dx) {
  auto nanat = [&](int yy, int xx) { const int64_t o = (int64_t)yy * nx + xx; const double v = u[o]; return sdf[o] == 0.0 || v != v; };
  const bool bad = nanat(y, x) || (y > 0 && nanat(y - 1, x)) || (y < ny - 1 && nanat(y + 1, x)) ||
                   (x > 0 && nanat(y, x - 1)) || (x < nx - 1 && nanat(y, x + 1));
  d_dy = 0.0; d_dx = 0.0;
  if (bad) return;
  const int ym = y > 0 ? y - 1 : y, yp = y < ny - 1 ? y + 1 : y;
  const int xm = x > 0 ? x - 1 : x, xp = x < nx - 1 ? x + 1 : x;
  const double hy = (y > 0 && y < ny - 1) ? 2.0 : 1.0, hx = (x > 0 && x < nx - 1) ? 2.0 : 1.0;
  d_dy = (u[(int64_t)yp * nx + x] - u[(int64_t)ym * nx + x]) / hy;
  d_dx = (u[(int64_t)y * nx + xp] - u[(int64_t)y * nx + xm]) / hx;
}

__device__ __forceinline__ double grad_term(const double* __restrict__ ux, const double* __restrict__ uy,
                                            const double* __restrict__ sdf, int y, int x, int ny, int nx, double scale) {
  double dUx_dy, dUx_dx, dUy_dy, dUy_dx;
  masked_grad(ux, sdf, y, x, ny, nx, dUx_dy, dUx_dx);
  masked_grad(uy, sdf, y, x, ny, nx, dUy_dy, dUy_dx);
  return (dUx_dx * dUx_dx + 2 * dUx_dy * dUy_dx + dUy_dy * dUy_dy) * scale;
}

__device__ __forceinline__ double block_sum(double v, double* red) {    // fixed tree order
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = FT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(FT) void psm_poisson_term_kernel(PsmFeatureArgs a) {
  __shared__ double red[FT];
  const int64_t pix = (int64_t)blockIdx.x * FT + threadIdx.x, n = (int64_t)a.ny * a.nx;
  double t = 0.0;
  if (pix < n) {
    const int y = (int)(pix / a.nx), x = (int)(pix - (int64_t)y * a.nx);
    t = grad_term(a.ux, a.uy, a.sdf, y, x, a.ny, a.nx, a.L * a.L / (a.U * a.U));   // (...) * L**2 / U**2, SMP:635
    a.term[pix] = t;
  }
  const double s1 = block_sum(t, red), s2 = block_sum(t * t, red);
  if (threadIdx.x == 0) { a.partial[2 * blockIdx.x] = s1; a.partial[2 * blockIdx.x + 1] = s2; }
}

__global__ __launch_bounds__(FT) void psm_poisson_grid_kernel(PsmFeatureArgs a) {
  __shared__ double red[FT];
  const int nwg = gridDim.x;
  double s1 = 0.0, s2 = 0.0;
  for (int w = threadIdx.x; w < nwg; w += FT) { s1 += a.partial[2 * w]; s2 += a.partial[2 * w + 1]; }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const int64_t n = (int64_t)a.ny * a.nx;
  const double mean = s1 / (double)n;
  double var = s2 / (double)n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double sd = sqrt(var);
  const double lo = mean - a.k * sd, hi = mean + a.k * sd;
  const int64_t pix = (int64_t)blockIdx.x * FT + threadIdx.x;
  if (pix >= n) return;
  const double t = a.term[pix];
  double sc;
  if (t < lo) sc = -1.0 - (t - lo) / lo;                    // SMP:50
  else if (t > hi) sc = 1.0 + (t - hi) / hi;                // SMP:52
  else sc = 2.0 * (t - lo) / (hi - lo) - 1.0;               // SMP:54
  double f0 = asinh(sc);                                    // SMP:60
  double f1 = a.dux[pix] / a.U, f2 = a.duy[pix] / a.U, f3 = a.sdf[pix];
  f0 = (f0 != f0) ? 0.0 : f0; f1 = (f1 != f1) ? 0.0 : f1;   // grid[np.isnan(grid)] = 0, SMP:704
  f2 = (f2 != f2) ? 0.0 : f2; f3 = (f3 != f3) ? 0.0 : f3;
  float4 o;
  o.x = (float)(f0 / a.max_abs[0]); o.y = (float)(f1 / a.max_abs[1]);
  o.z = (float)(f2 / a.max_abs[2]); o.w = (float)(f3 / a.max_abs[3]);
  reinterpret_cast<float4*>(a.grid)[pix] = o;
}
}  // namespace

hipError_t psm_launch_poisson_features(const PsmFeatureArgs& a, hipStream_t st) {
  const int64_t n = (int64_t)a.ny * a.nx;
  const unsigned nwg = (unsigned)((n + FT - 1) / FT);
  hipLaunchKernelGGL(psm_poisson_term_kernel, dim3(nwg), dim3(FT), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(psm_poisson_grid_kernel, dim3(nwg), dim3(FT), 0, st, a);
  return hipGetLastError();
}
