// psm_mesh.h -- launchers of the mesh <-> grid kernels (see psm_mesh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct PsmToGridArgs {
  const double* cells;          // [N,5] Ux,Uy,Cx,Cy,p
  const double* umax;           // device scalar, or nullptr: umax_val (computed by the host while the H2D copy runs)
  double umax_val;
  const double* umax_partials;  // large meshes: per-workgroup partial maxima of psm_umax_partial_kernel (reduced here by every
  int n_partials;               // workgroup; workgroup 0 also stores the result to umax_out for psm_to_mesh_kernel), or nullptr
  double* umax_out;
  const int32_t* vtx;           // [n_grid,3] mesh->grid simplices (interp_weights, PM:52-62)
  const double* wts;            // [n_grid,3]
  const int32_t* src_of_cell;   // [n_grid] last grid point scattered into each cell (NumPy fancy assignment order), -1 none
  const double* sdf;            // [n_grid] sdfunct
  float* grid;                  // [n_grid][c_in]
  int64_t n_grid;
  double max_abs_ux, max_abs_uy, sdf_scale;
  int c_in, fill;               // fill: 1 = interpolate_fill (NaN where a weight is negative), 0 = interpolate
};

struct PsmToMeshArgs {
  const double* cells;          // [N,5]
  const double* umax;           // device scalar, or nullptr: umax_val
  double umax_val;
  const int32_t* vtx;           // [N,3] grid->mesh simplices
  const double* wts;            // [N,3]
  const int32_t* cell_of_point; // [n_grid] flat cell index of indices[point]
  const float* field;           // [n_grid][c_out]
  const uint8_t* near_wall;     // [N] interpolated SDF < threshold (PM:492-494), constant per geometry
  double* p_out;                // [N]
  int64_t n_cells;
  double max_abs_p;
  int c_out;
};

hipError_t psm_launch_umax(const double* cells, int64_t n, double* umax, hipStream_t st);
// parallel form for large meshes: partials[0 .. *n_partials) (capacity 256), reduced by psm_to_grid_kernel
hipError_t psm_launch_umax_partial(const double* cells, int64_t n, double* partials, int* n_partials, hipStream_t st);
// registered caller buffers: host cells -> device copy + partial maxima (capacity 256) in one kernel, see psm_mesh.hip
hipError_t psm_launch_stage_cells(const double* host_cells, double* cells, int64_t n, double* partials, int* n_partials, hipStream_t st);
hipError_t psm_launch_to_grid(const PsmToGridArgs& a, hipStream_t st);
hipError_t psm_launch_to_mesh(const PsmToMeshArgs& a, hipStream_t st);
hipError_t psm_launch_interp_to_grid(const double* values, int k, const int32_t* vtx, const double* wts, const int32_t* src_of_cell,
                                     int fill, double* out, int64_t n_grid, hipStream_t st);
// one pass of the separable Gaussian filter (axis 0 = rows direction, 1 = columns)
hipError_t psm_launch_gauss1d(const float* in, float* out, int ny, int nx, int axis, int radius, const float* wts, hipStream_t st);

// label blocks [B][S*S*c_out] with the per-block flow-cell mean removed (SM_call.py:487-488, UGP:509-511)
hipError_t psm_launch_label_blocks(const float* grid, const float* labels, const int32_t* blk_y0x0, float* out, int B, int S,
                                   int c_in, int c_out, int sdf_ch, int Nx, hipStream_t st);
// compute_in_block_error (utils.py:210-243): per-block partial sums [B][8] doubles, see psm_mesh.hip
hipError_t psm_launch_block_error(const float* grid, const float* pred, const float* label_blocks, const float* row_scale,
                                  const int32_t* blk_y0x0, double* part, int B, int S, int c_in, int c_out, int sdf_ch, int Nx, hipStream_t st);

// ---- U_to_gradP integration (UGP:371-416, 592-628)
constexpr int PSM_INTEG_MAX_FIX = 4;   // distinct indices the "reset" quirk may touch per row
struct PsmIntegArgs {
  const float* gradp;        // [ny][nx][2]
  const int2* fixups;        // [max(cy, ny-cy)][PSM_INTEG_MAX_FIX] (v, u), v = -1: unused
  const int2* pairs;         // [npair[0] + npair[1]] (row in left block, row in right block), global rows
  int npair[2];
  float* rxl; float* rxr;    // [ny][cx], [ny][nx-cx+1]
  float* yl; float* yr;      // [ny]
  float* corr;               // [2]
  float* p_out;              // [ny][nx]
  int ny, nx, cy, cx;
  float dx, dy;
};
hipError_t psm_launch_integrate(const PsmIntegArgs& a, hipStream_t st);

// ---- pressureSM_Poisson input features (SMP:588-711), see psm_features.hip
struct PsmFeatureArgs {
  const double *ux, *uy, *dux, *duy, *sdf;   // [ny][nx] float64 (dimensional grids, zero outside the flow; raw SDF)
  double* term;                              // [ny][nx] scratch: the Poisson source term
  double* partial;                           // [2 * workgroups] (sum, sum of squares)
  float* grid;                               // [ny][nx][4] float32 NHWC
  int ny, nx;
  double L, U, k;
  double max_abs[4];                         // Poisson_term_1, delta_Ux, delta_Uy, dist
};
hipError_t psm_launch_poisson_features(const PsmFeatureArgs& a, hipStream_t st);
