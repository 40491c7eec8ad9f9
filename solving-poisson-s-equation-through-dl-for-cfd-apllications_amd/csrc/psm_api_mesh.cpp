// psm_api_mesh.cpp -- C-ABI of libpsm_hip.so (include/psm.h): solver boundary (mesh <-> grid) and evaluator helpers.  See psm_handle.h for the map of the five files.
#include "psm_handle.h"

namespace psm_impl {

// psm_solve on registered, mapped caller buffers: every device-side step of the call, in stream order (captured once)
int mesh_sequence(psm_handle* h, int64_t n, hipStream_t st) {
  int n_partials = 0;
  HIPCHK(h, psm_launch_stage_cells(h->pinned_cells_dev, h->d_cells, n, h->d_umax_part, &n_partials, st));
  PsmToGridArgs ga{};
  ga.cells = h->d_cells; ga.umax = nullptr; ga.umax_val = 0.0;
  ga.umax_partials = h->d_umax_part; ga.n_partials = n_partials; ga.umax_out = h->d_umax;
  ga.vtx = h->d_vtx_m2g; ga.wts = h->d_wts_m2g; ga.src_of_cell = h->d_src_of_cell;
  ga.sdf = h->d_sdf; ga.grid = h->d_grid_stage; ga.n_grid = (int64_t)h->Ny * h->Nx;
  ga.max_abs_ux = h->maxs[0]; ga.max_abs_uy = h->maxs[1]; ga.sdf_scale = h->normalise_sdf ? 1.0 / h->maxs[2] : 1.0;
  ga.c_in = h->cfg.c_in; ga.fill = h->fill_input;
  HIPCHK(h, psm_launch_to_grid(ga, st));
  h->in_mesh_solve = true;
  int rc = launch_all(h, h->ws0, h->d_grid_stage, 1, h->d_fields_stage, h->d_ones, st, nullptr);
  h->in_mesh_solve = false;
  if (rc) return rc;
  PsmToMeshArgs ma{};
  ma.cells = h->d_cells; ma.umax = h->d_umax; ma.umax_val = 0.0; ma.vtx = h->d_vtx_g2m; ma.wts = h->d_wts_g2m; ma.cell_of_point = h->d_cell_of_point;
  ma.field = h->d_fields_stage; ma.near_wall = h->d_near_wall; ma.p_out = h->pinned_p_dev; ma.n_cells = n; ma.max_abs_p = h->maxs[3];
  ma.c_out = h->cfg.c_out;
  HIPCHK(h, psm_launch_to_mesh(ma, st));
  return PSM_OK;
}

}  // namespace psm_impl

// ============================================================================
extern "C" {


int psm_reassemble(psm_handle* h, const float* grid, const float* block_pred, float* fields) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!grid || !block_pred || !fields) return fail(h, PSM_ERR_ARG, "null buffer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t npix = (size_t)h->Ny * h->Nx;
  HIPCHK(h, hipStreamSynchronize(st));                    // the staging buffers are free; caller memory goes through the bounce buffer
  HIPCHK(h, psm_copy_h2d(h->d_grid_stage, grid, npix * h->cfg.c_in * sizeof(float)));
  HIPCHK(h, psm_copy_h2d(h->ws0.d_pred, block_pred, (size_t)h->B * h->K_out * sizeof(float)));
  PsmStripArgs sa{};
  sa.pred = h->ws0.d_pred; sa.grid = h->d_grid_stage; sa.strips = h->d_strips; sa.blk_y0x0 = h->d_blk; sa.spart = h->ws0.d_spart; sa.colpart = h->ws0.d_colpart; sa.NS = h->plan.cp.NS; sa.n_bands = h->n_bands;
  sa.B = h->B; sa.S = h->S; sa.c_in = h->cfg.c_in; sa.c_out = h->cfg.c_out;
  sa.sdf_ch = h->cfg.sdf_channel; sa.Ny = h->Ny; sa.Nx = h->Nx;
  HIPCHK(h, psm_launch_strips(sa, 1, st));
  PsmChainArgs ca{};
  ca.cp = h->plan.cp; ca.blocks = h->d_blocks; ca.spart = h->ws0.d_spart; ca.colpart = h->ws0.d_colpart; ca.n_bands = h->n_bands; ca.pred = h->ws0.d_pred; ca.owner = h->d_owner;
  ca.shiftA = h->d_shiftA; ca.shiftB = h->d_shiftB; ca.shiftOwnA = h->d_shiftOwnA; ca.shiftOwnB = h->d_shiftOwnB; ca.shiftW = h->d_shiftW;
  for (int f = 0; f < 2; ++f) ca.shiftL[f] = (int)h->plan.shiftA[f].size();
  ca.Lmax = h->Lmax; ca.offs = h->ws0.d_offs; ca.shift = h->ws0.d_shift; ca.n_strips = h->n_strips; ca.c_out = h->cfg.c_out; ca.stamps = h->d_stamps;
  HIPCHK(h, psm_launch_chain(ca, 1, st));
  PsmPasteArgs pa{h->ws0.d_pred, h->d_owner, h->ws0.d_offs, h->ws0.d_shift, h->d_fields_stage, h->B, h->S, h->cfg.c_out, h->Ny * h->Nx};
  HIPCHK(h, psm_launch_paste(pa, 1, st));
  HIPCHK(h, wait_stream(st));
  HIPCHK(h, psm_copy_d2h(fields, h->d_fields_stage, npix * h->cfg.c_out * sizeof(float)));
  h->last_cases = 1;
  return PSM_OK;
}


int psm_label_blocks(psm_handle* h, const float* grid, const float* labels, float* blocks_out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!grid || !labels || !blocks_out) return fail(h, PSM_ERR_ARG, "null buffer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gb = npix * h->cfg.c_in * sizeof(float), lb = npix * h->cfg.c_out * sizeof(float), ob = (size_t)h->B * h->K_out * sizeof(float);
  int rc;
  if ((rc = scratch_reserve(h, carve_size({gb, lb, ob}), carve_size({gb, lb, ob})))) return rc;
  Carver cd{(char*)h->scr_dev}, cp{(char*)h->scr_pin};
  float* d_g = cd.take<float>(npix * h->cfg.c_in); float* d_l = cd.take<float>(npix * h->cfg.c_out); float* d_o = cd.take<float>((size_t)h->B * h->K_out);
  float* p_g = cp.take<float>(npix * h->cfg.c_in); float* p_l = cp.take<float>(npix * h->cfg.c_out); float* p_o = cp.take<float>((size_t)h->B * h->K_out);
  memcpy(p_g, grid, gb); memcpy(p_l, labels, lb);
  hipError_t e = hipMemcpyAsync(d_g, p_g, gb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_l, p_l, lb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = psm_launch_label_blocks(d_g, d_l, h->d_blk, d_o, h->B, h->S, h->cfg.c_in, h->cfg.c_out, h->cfg.sdf_channel, h->Nx, st);
  if (e == hipSuccess) e = hipMemcpyAsync(p_o, d_o, ob, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = wait_stream(st);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("label blocks: ") + hipGetErrorString(e));
  memcpy(blocks_out, p_o, ob);
  return PSM_OK;
}


int psm_block_error(psm_handle* h, const float* grid, const float* labels, double* out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned || h->last_cases < 1) return fail(h, PSM_ERR_STATE, "no solve has run yet");
  // The network output it decodes lives in the handle's own workspace.  A solve through the asynchronous ring
  // (psm_submit_grid*, psm_ring_*, psm_bench_host) ran on a ring slot's workspace and left an OLDER solve here.
  if (!h->last_on_ws0)
    return fail(h, PSM_ERR_STATE, "psm_block_error follows a synchronous solve (psm_solve_grid / psm_solve_grid_device / psm_solve); the last solve ran on the ring");
  if (!grid || !labels || !out) return fail(h, PSM_ERR_ARG, "null buffer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipDeviceSynchronize());                        // the solve may have run on the caller's stream
  hipStream_t st = h->stream;
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gb = npix * h->cfg.c_in * sizeof(float), lb = npix * h->cfg.c_out * sizeof(float), ob = (size_t)h->B * h->K_out * sizeof(float);
  const size_t pb = (size_t)h->B * 8 * sizeof(double);
  int rc;
  if ((rc = scratch_reserve(h, carve_size({gb, lb, ob, pb}), carve_size({gb, lb, pb})))) return rc;
  Carver cd{(char*)h->scr_dev}, cp{(char*)h->scr_pin};
  float* d_g = cd.take<float>(npix * h->cfg.c_in); float* d_l = cd.take<float>(npix * h->cfg.c_out); float* d_o = cd.take<float>((size_t)h->B * h->K_out);
  double* d_p = cd.take<double>((size_t)h->B * 8);
  float* p_g = cp.take<float>(npix * h->cfg.c_in); float* p_l = cp.take<float>(npix * h->cfg.c_out); double* p_p = cp.take<double>((size_t)h->B * 8);
  memcpy(p_g, grid, gb); memcpy(p_l, labels, lb);
  // the decoded blocks of the last solve (case 0): on the geometry-bound path they were never stored -- decode its network output again
  const int M = h->B, Mpad = round_up(M, 32);
  const float* scale = h->last_row_scale ? h->last_row_scale : h->d_ones;
  PsmDecodeArgs de{};
  de.res = h->ws0.d_res; de.ld_res = h->ld_out; de.bpack = h->d_bpack_out; de.mean = h->d_mean_out;
  de.row_scale = scale; de.pred = h->ws0.d_pred; de.M = M; de.Mpad = Mpad; de.Gd = h->Gd; de.n_coltiles = h->n_coltiles; de.K_out = h->K_out;
  const bool bf16 = h->cfg.precision == PSM_PRECISION_BF16;
  hipError_t e = bf16 ? psm_launch_decode_bf16(de, st) : psm_launch_decode(de, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_g, p_g, gb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_l, p_l, lb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = psm_launch_label_blocks(d_g, d_l, h->d_blk, d_o, h->B, h->S, h->cfg.c_in, h->cfg.c_out, h->cfg.sdf_channel, h->Nx, st);
  if (e == hipSuccess) e = psm_launch_block_error(d_g, h->ws0.d_pred, d_o, scale, h->d_blk, d_p, h->B, h->S, h->cfg.c_in, h->cfg.c_out, h->cfg.sdf_channel, h->Nx, st);
  if (e == hipSuccess) e = hipMemcpyAsync(p_p, d_p, pb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = wait_stream(st);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("block error: ") + hipGetErrorString(e));
  double n = 0, s1 = 0, s2 = 0, tmin = INFINITY, tmax = -INFINITY, pmin = INFINITY, pmax = -INFINITY, tnan = 0;
  for (int b = 0; b < h->B; ++b) {
    const double* q = p_p + (size_t)b * 8;
    n += q[0]; s1 += q[1]; s2 += q[2]; tnan += q[7];
    tmin = std::min(tmin, q[3]); tmax = std::max(tmax, q[4]); pmin = std::min(pmin, q[5]); pmax = std::max(pmax, q[6]);
  }
  const double norm = tnan > 0 ? NAN : tmax - tmin;        // np.max / np.min propagate a NaN label
  out[0] = s1 / n / norm;                                   // pred_minus_true_block (utils.py:241)
  out[1] = s2 / n / (norm * norm);                          // pred_minus_true_squared_block (utils.py:242)
  out[2] = norm; out[3] = pmax - pmin; out[4] = n;
  return PSM_OK;
}


int psm_set_geometry(psm_handle* h, int64_t n_cells, int32_t ny, int32_t nx, const int32_t* vtx_m2g, const double* wts_m2g,
                     const int32_t* indices, const double* sdfunct, const int32_t* vtx_g2m, const double* wts_g2m,
                     const double* maxs, int32_t normalise_sdf, int32_t fill_input, double wall_threshold) {
  if (!h) return PSM_ERR_ARG;
  if (!vtx_m2g || !wts_m2g || !indices || !sdfunct || !maxs) return fail(h, PSM_ERR_ARG, "null geometry table");
  if ((vtx_g2m == nullptr) != (wts_g2m == nullptr)) return fail(h, PSM_ERR_ARG, "vtx_g2m and wts_g2m go together");
  const bool g2m = vtx_g2m != nullptr;
  if (n_cells < 1 || n_cells > (int64_t)1 << 30) return fail(h, PSM_ERR_ARG, "bad cell count");
  const int64_t ng = (int64_t)ny * nx;
  for (int64_t t = 0; t < ng; ++t) {
    for (int j = 0; j < 3; ++j)
      if (vtx_m2g[t * 3 + j] < 0 || vtx_m2g[t * 3 + j] >= n_cells) return fail(h, PSM_ERR_ARG, "mesh->grid vertex index out of range");
    if (indices[t * 2] < 0 || indices[t * 2] >= ny || indices[t * 2 + 1] < 0 || indices[t * 2 + 1] >= nx)
      return fail(h, PSM_ERR_ARG, "indices outside the grid");
  }
  for (int64_t n = 0; g2m && n < n_cells; ++n)
    for (int j = 0; j < 3; ++j)
      if (vtx_g2m[n * 3 + j] < 0 || vtx_g2m[n * 3 + j] >= ng) return fail(h, PSM_ERR_ARG, "grid->mesh vertex index out of range");
  int rc = psm_plan_grid(h, ny, nx);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  free_geometry(h);
  h->n_cells = n_cells;
  for (int k = 0; k < 4; ++k) h->maxs[k] = maxs[k];
  h->normalise_sdf = normalise_sdf; h->fill_input = fill_input;
  // NumPy fancy assignment grid[...][tuple(indices.T)] = values writes in point order: last wins
  std::vector<int32_t> src(ng, -1), cop(ng);
  for (int64_t t = 0; t < ng; ++t) {
    const int64_t cell = (int64_t)indices[t * 2] * nx + indices[t * 2 + 1];
    src[cell] = (int32_t)t;
    cop[t] = (int32_t)cell;
  }
  // sdf_mesh = interpolate_fill(sdfunct.flatten(), vert_NPtoOF, weights_NPtoOF) < threshold  (PM:492-494)
  std::vector<uint8_t> nw(n_cells, 0);
  for (int64_t n = 0; g2m && n < n_cells; ++n) {
    double acc = 0.0; bool neg = false;
    for (int j = 0; j < 3; ++j) { acc += sdfunct[vtx_g2m[n * 3 + j]] * wts_g2m[n * 3 + j]; neg = neg || wts_g2m[n * 3 + j] < 0.0; }
    nw[n] = (!neg && acc < wall_threshold) ? 1 : 0;     // NaN (fill) compares false
  }
  std::vector<int32_t> v1(vtx_m2g, vtx_m2g + ng * 3), v2;
  std::vector<double> w1(wts_m2g, wts_m2g + ng * 3), w2, sd(sdfunct, sdfunct + ng);
  if (g2m) { v2.assign(vtx_g2m, vtx_g2m + n_cells * 3); w2.assign(wts_g2m, wts_g2m + n_cells * 3); }
  else { v2.assign((size_t)n_cells * 3, 0); w2.assign((size_t)n_cells * 3, 0.0); }
  h->have_g2m = g2m;
  if ((rc = dev_upload(h, &h->d_vtx_m2g, v1))) return rc;
  if ((rc = dev_upload(h, &h->d_wts_m2g, w1))) return rc;
  if ((rc = dev_upload(h, &h->d_src_of_cell, src))) return rc;
  if ((rc = dev_upload(h, &h->d_cell_of_point, cop))) return rc;
  if ((rc = dev_upload(h, &h->d_sdf, sd))) return rc;
  if ((rc = dev_upload(h, &h->d_vtx_g2m, v2))) return rc;
  if ((rc = dev_upload(h, &h->d_wts_g2m, w2))) return rc;
  if ((rc = dev_upload(h, &h->d_near_wall, nw))) return rc;
  if ((rc = dev_alloc(h, &h->d_cells, (size_t)n_cells * 5))) return rc;
  if ((rc = dev_alloc(h, &h->d_p, (size_t)n_cells))) return rc;
  if ((rc = dev_alloc(h, &h->d_umax, (size_t)1))) return rc;
  if ((rc = dev_alloc(h, &h->d_umax_part, (size_t)256))) return rc;
  HIPCHK(h, hipHostMalloc((void**)&h->h_cells, (size_t)n_cells * 5 * sizeof(double), hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc((void**)&h->h_p, (size_t)n_cells * sizeof(double), hipHostMallocDefault));
  h->have_geometry = true;
  // The mesh entry builds its grid from THIS sdfunct at every step, so the geometry of psm_solve is fixed from here
  // on: bind it (scope: psm_solve only -- grid-native solves on the same handle stay general until psm_bind_geometry).
  if (h->cfg.c_in == 3 && h->cfg.sdf_channel == 2 && g2m && getenv("PSM_NO_BIND") == nullptr) {
    std::vector<float> g((size_t)ng * 3, 0.f);
    const double sc = normalise_sdf ? 1.0 / maxs[2] : 1.0;
    for (int64_t t = 0; t < ng; ++t) {
      const double sdv = sdfunct[t] * sc;                    // the SDF channel exactly as psm_to_grid_kernel writes it
      g[(size_t)t * 3 + 2] = (sdv != sdv) ? 0.f : (float)sdv;
    }
    HIPCHK(h, psm_copy_h2d(h->d_grid_stage, g.data(), g.size() * sizeof(float)));
    rc = bind_geometry_device(h, h->d_grid_stage);
    if (rc == PSM_OK) h->bound_scope = 1;
    else if (rc == PSM_ERR_UNSUPPORTED) h->err.clear();      // configuration outside the fused path: general path
    else return rc;
  }
  return PSM_OK;
}


int psm_set_case(psm_handle* h, const double* maxs, double delta, int32_t every, double wall_threshold) {
  if (!h) return PSM_ERR_ARG;
  if (!maxs || !(delta > 0.0) || every < 1 || !(wall_threshold >= 0.0)) return fail(h, PSM_ERR_ARG, "bad case constants");
  for (int k = 0; k < 4; ++k) {
    if (!(maxs[k] != 0.0)) return fail(h, PSM_ERR_ARG, "maxs must be non-zero");
    h->case_maxs[k] = maxs[k];
  }
  h->case_delta = delta; h->case_every = every; h->case_wall = wall_threshold;
  return PSM_OK;
}


int psm_init_geometry(psm_handle* h, const double* cells, int64_t n, const double* top, int64_t n_top, const double* obst,
                      int64_t n_obst, int32_t rank) {
  (void)rank;
  if (!h) return PSM_ERR_ARG;
  if (!cells || !top || !obst) return fail(h, PSM_ERR_ARG, "null buffer");
  int32_t ny = 0, nx = 0;
  if (psm_geometry_shape(cells, n, h->case_delta, &ny, &nx, nullptr) != PSM_OK) return fail(h, PSM_ERR_ARG, psm_geometry_last_error());
  const size_t ng = (size_t)ny * nx;
  std::vector<int32_t> v1(ng * 3), idx(ng * 2), v2((size_t)n * 3);
  std::vector<double> w1(ng * 3), sdf(ng), w2((size_t)n * 3);
  int rc = psm_geometry_build(cells, n, top, n_top, obst, n_obst, h->case_delta, h->case_every, v1.data(), w1.data(), idx.data(),
                              sdf.data(), v2.data(), w2.data());
  if (rc) return fail(h, rc, psm_geometry_last_error());
  return psm_set_geometry(h, n, ny, nx, v1.data(), w1.data(), idx.data(), sdf.data(), v2.data(), w2.data(), h->case_maxs, 0, 0, h->case_wall);
}


int psm_solve_begin(psm_handle* h, const double* cells, int64_t n, int32_t rank, double* p_out) {
  (void)rank;
  if (!h) return PSM_ERR_ARG;
  if (h->mesh_inflight) return fail(h, PSM_ERR_STATE, "a psm_solve_begin is already in flight on this handle: call psm_solve_end first");
  if (!h->have_geometry || !h->planned)
    return fail(h, PSM_ERR_STATE, "psm_set_geometry has not been called (or the plan it belonged to was dropped by a later psm_set_* / psm_plan_grid)");
  if (h->cfg.c_in != 3 || h->cfg.c_out != 1) return fail(h, PSM_ERR_UNSUPPORTED, "the mesh entry needs c_in == 3 and c_out == 1 (python_module.py:288-292)");
  if (!h->have_g2m) return fail(h, PSM_ERR_STATE, "psm_set_geometry was called without the grid->mesh tables");
  if (!cells || !p_out) return fail(h, PSM_ERR_ARG, "null buffer");
  if (n != h->n_cells) return fail(h, PSM_ERR_ARG, "cell count differs from the geometry");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  { int rc0 = ensure_encode_aux(h, 1); if (rc0) return rc0; }   // many-block cases (the shipped 104-block shape): the M-tiled encode's split basis, once, outside any capture
  // Both arrays registered (psm_pin_buffers) and mapped: the whole call is ONE hipGraph replay -- psm_stage_cells_kernel reads
  // the cells over PCIe and takes the partial maxima of U_max on the way (no DMA-engine copy, no host pass, U_max never leaves
  // the device: to_grid reduces the partials and hands the scalar to to_mesh through d_umax), to_grid, the kernels of the
  // solve, to_mesh storing p straight into the caller's array.
  // PSM_MESH_GRAPH: 0 = the separate submissions below (DMA copy, host U_max), 1 = one graph replay, 2 = the same sequence as
  // plain launches (measured default)
  static const int mesh_mode = getenv("PSM_MESH_GRAPH") ? atoi(getenv("PSM_MESH_GRAPH")) : 2;
  static const int64_t stage_max = getenv("PSM_MESH_STAGE_MAX") ? atoll(getenv("PSM_MESH_STAGE_MAX")) : PSM_MESH_STAGE_MAX_DEFAULT;
  if (mesh_mode != 0 && h->timed_kernel < 0 && n <= stage_max && cells == h->pinned_cells && h->pinned_cells_dev && p_out == h->pinned_p && h->pinned_p_dev) {
    h->last_cases = 1;
    if (mesh_mode == 2) {
      int rc = mesh_sequence(h, n, st);
      if (rc) return rc;
    } else {
      if (!h->mesh_graph) {
        hipGraph_t graph = nullptr;
        HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int rc = mesh_sequence(h, n, st);
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); return fail(h, PSM_ERR_HIP, std::string("psm_solve capture: ") + hipGetErrorString(e)); }
        e = hipGraphInstantiate(&h->mesh_graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { h->mesh_graph = nullptr; return fail(h, PSM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
      }
      HIPCHK(h, hipGraphLaunch(h->mesh_graph, st));
    }
    h->mesh_copy_out = nullptr;
    h->mesh_inflight = true;
    return PSM_OK;
  }
  if (cells == h->pinned_cells) {            // registered by the caller: DMA straight from its buffer
    HIPCHK(h, hipMemcpyAsync(h->d_cells, cells, (size_t)n * 5 * sizeof(double), hipMemcpyHostToDevice, st));
  } else {
    memcpy(h->h_cells, cells, (size_t)n * 5 * sizeof(double));
    HIPCHK(h, hipMemcpyAsync(h->d_cells, h->h_cells, (size_t)n * 5 * sizeof(double), hipMemcpyHostToDevice, st));
  }
  // U_max = max sqrt(Ux^2 + Uy^2) (PM:270) on the host while the copy above is in flight: sqrt is monotonic and
  // correctly rounded on both sides, so sqrt(max(Ux^2 + Uy^2)) is the kernel's value bit for bit (NaN propagates
  // like np.max); one launch less.  PSM_DEVICE_UMAX=1 keeps the device reduction.
  // Large meshes (the host pass would take longer than the copy it hides under): parallel device reduction, whose
  // per-workgroup maxima every psm_to_grid workgroup folds itself.
  static const bool dev_umax_env = getenv("PSM_DEVICE_UMAX") != nullptr;
  const bool big = n > 32768;
  const bool dev_umax = dev_umax_env && !big;
  double umax_val = 0.0;
  int n_partials = 0;
  if (big) {
    HIPCHK(h, psm_launch_umax_partial(h->d_cells, n, h->d_umax_part, &n_partials, st));
  } else if (dev_umax) {
    HIPCHK(h, psm_launch_umax(h->d_cells, n, h->d_umax, st));
  } else {
    double m2 = 0.0; bool nan = false;
    for (int64_t i = 0; i < n; ++i) {
      const double ux = cells[i * 5], uy = cells[i * 5 + 1];
      const double v = ux * ux + uy * uy;
      nan = nan || (v != v);
      m2 = v > m2 ? v : m2;
    }
    umax_val = nan ? std::nan("") : std::sqrt(m2);
  }
  PsmToGridArgs ga{};
  ga.cells = h->d_cells; ga.umax = dev_umax ? h->d_umax : nullptr; ga.umax_val = umax_val;
  if (big) { ga.umax_partials = h->d_umax_part; ga.n_partials = n_partials; ga.umax_out = h->d_umax; } ga.vtx = h->d_vtx_m2g; ga.wts = h->d_wts_m2g; ga.src_of_cell = h->d_src_of_cell;
  ga.sdf = h->d_sdf; ga.grid = h->d_grid_stage; ga.n_grid = (int64_t)h->Ny * h->Nx;
  ga.max_abs_ux = h->maxs[0]; ga.max_abs_uy = h->maxs[1]; ga.sdf_scale = h->normalise_sdf ? 1.0 / h->maxs[2] : 1.0;
  ga.c_in = h->cfg.c_in; ga.fill = h->fill_input;
  HIPCHK(h, psm_launch_to_grid(ga, st));
  h->in_mesh_solve = true;
  int rc = solve_device(h, h->d_grid_stage, 1, nullptr, h->d_fields_stage, st, nullptr);
  h->in_mesh_solve = false;
  if (rc) return rc;
  PsmToMeshArgs ma{};
  ma.cells = h->d_cells; ma.umax = (dev_umax || big) ? h->d_umax : nullptr; ma.umax_val = umax_val; ma.vtx = h->d_vtx_g2m; ma.wts = h->d_wts_g2m; ma.cell_of_point = h->d_cell_of_point;
  ma.field = h->d_fields_stage; ma.near_wall = h->d_near_wall; ma.p_out = h->d_p; ma.n_cells = n; ma.max_abs_p = h->maxs[3];
  ma.c_out = h->cfg.c_out;
  const bool direct = p_out == h->pinned_p && h->pinned_p_dev != nullptr;
  if (direct) ma.p_out = h->pinned_p_dev;                 // 8 bytes per cell over PCIe from the kernel itself: no D2H copy
  HIPCHK(h, psm_launch_to_mesh(ma, st));
  if (p_out == h->pinned_p) {
    if (!direct) HIPCHK(h, hipMemcpyAsync(p_out, h->d_p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    h->mesh_copy_out = nullptr;
  } else {
    HIPCHK(h, hipMemcpyAsync(h->h_p, h->d_p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    h->mesh_copy_out = p_out;
  }
  h->mesh_inflight = true;
  return PSM_OK;
}


int psm_solve_end(psm_handle* h) {
  if (!h) return PSM_ERR_ARG;
  if (!h->mesh_inflight) return fail(h, PSM_ERR_STATE, "no psm_solve_begin in flight");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->mesh_inflight = false;
  HIPCHK(h, wait_stream(h->stream));
  if (h->mesh_copy_out) memcpy(h->mesh_copy_out, h->h_p, (size_t)h->n_cells * sizeof(double));
  return PSM_OK;
}


int psm_solve(psm_handle* h, const double* cells, int64_t n, int32_t rank, double* p_out) {
  int rc = psm_solve_begin(h, cells, n, rank, p_out);
  return rc ? rc : psm_solve_end(h);
}



int psm_pin_buffers(psm_handle* h, const double* cells, double* p_out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->have_geometry) return fail(h, PSM_ERR_STATE, "psm_set_geometry has not been called");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unpin_buffers(h);
  if (cells) {
    hipError_t e = hipHostRegister((void*)cells, (size_t)h->n_cells * 5 * sizeof(double), hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, PSM_ERR_HIP, std::string("hipHostRegister(cells): ") + hipGetErrorString(e)); }
    h->pinned_cells = cells;
    void* dc = nullptr;                                   // mapped address: lets psm_stage_cells_kernel read the cells from the host array
    if (hipHostGetDevicePointer(&dc, (void*)cells, 0) == hipSuccess && getenv("PSM_NO_DIRECT_IN") == nullptr) h->pinned_cells_dev = (const double*)dc;
    else (void)hipGetLastError();
  }
  if (p_out) {
    hipError_t e = hipHostRegister((void*)p_out, (size_t)h->n_cells * sizeof(double), hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); unpin_buffers(h); return fail(h, PSM_ERR_HIP, std::string("hipHostRegister(p_out): ") + hipGetErrorString(e)); }
    h->pinned_p = p_out;
    void* dp = nullptr;                                   // mapped address: lets psm_to_mesh_kernel store p into the host array
    if (hipHostGetDevicePointer(&dp, (void*)p_out, 0) == hipSuccess && getenv("PSM_NO_DIRECT_OUT") == nullptr) h->pinned_p_dev = (double*)dp;
    else (void)hipGetLastError();
  }
  return PSM_OK;
}


int psm_unpin_buffers(psm_handle* h) {
  if (!h) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unpin_buffers(h);
  return PSM_OK;
}


int psm_gaussian_filter(psm_handle* h, const float* in, int32_t ny, int32_t nx, double sigma_y, double sigma_x, float* out) {
  if (!h) return PSM_ERR_ARG;
  if (!in || !out || ny < 1 || nx < 1 || (int64_t)ny * nx > ((int64_t)1 << 28)) return fail(h, PSM_ERR_ARG, "bad field");
  if (!(sigma_y > 0.0) || !(sigma_x > 0.0) || sigma_y > 1e4 || sigma_x > 1e4) return fail(h, PSM_ERR_ARG, "sigma must be positive");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t n = (size_t)ny * nx;
  float *d_a = nullptr, *d_b = nullptr, *d_w = nullptr;
  int rc = PSM_OK;
  auto weights = [](double sigma, std::vector<float>& w) {      // scipy.ndimage._gaussian_kernel1d, order 0
    const int r = (int)(4.0 * sigma + 0.5);
    std::vector<double> p(2 * r + 1);
    double sum = 0.0;
    for (int x = -r; x <= r; ++x) { p[x + r] = std::exp(-0.5 / (sigma * sigma) * (double)x * (double)x); sum += p[x + r]; }
    w.resize(2 * r + 1);
    for (int k = 0; k < 2 * r + 1; ++k) w[k] = (float)(p[k] / sum);
    return r;
  };
  std::vector<float> wy, wx;
  const int ry = weights(sigma_y, wy), rx = weights(sigma_x, wx);
  std::vector<float> wall(wy);
  wall.insert(wall.end(), wx.begin(), wx.end());
  const size_t nb = n * sizeof(float), wb = wall.size() * sizeof(float);
  if ((rc = scratch_reserve(h, carve_size({nb, nb, wb}), carve_size({nb, wb})))) return rc;
  Carver cd{(char*)h->scr_dev}, cp{(char*)h->scr_pin};
  d_a = cd.take<float>(n); d_b = cd.take<float>(n); d_w = cd.take<float>(wall.size());
  float* p_io = cp.take<float>(n); float* p_w = cp.take<float>(wall.size());
  memcpy(p_io, in, nb); memcpy(p_w, wall.data(), wb);
  hipError_t e = hipMemcpyAsync(d_a, p_io, nb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_w, p_w, wb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = psm_launch_gauss1d(d_a, d_b, ny, nx, 0, ry, d_w, st);
  if (e == hipSuccess) e = psm_launch_gauss1d(d_b, d_a, ny, nx, 1, rx, d_w + wy.size(), st);
  if (e == hipSuccess) e = hipMemcpyAsync(p_io, d_a, nb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = wait_stream(st);
  if (e == hipSuccess) memcpy(out, p_io, nb);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("gaussian filter: ") + hipGetErrorString(e));
  return PSM_OK;
}


int psm_mesh_to_grid(psm_handle* h, const double* values, int64_t n, int32_t k, int32_t fill, double* grid_out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->have_geometry) return fail(h, PSM_ERR_STATE, "psm_set_geometry has not been called");
  if (!values || !grid_out) return fail(h, PSM_ERR_ARG, "null buffer");
  if (n != h->n_cells) return fail(h, PSM_ERR_ARG, "cell count differs from the geometry");
  if (k < 1 || k > 16) return fail(h, PSM_ERR_ARG, "1..16 columns");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t ng = (size_t)h->Ny * h->Nx;
  int rc;
  const size_t vb = (size_t)n * k * sizeof(double), ob = ng * k * sizeof(double);
  if ((rc = scratch_reserve(h, carve_size({vb, ob}), carve_size({vb, ob})))) return rc;
  Carver cd{(char*)h->scr_dev}, cp{(char*)h->scr_pin};
  double* d_v = cd.take<double>((size_t)n * k); double* d_o = cd.take<double>(ng * k);
  double* p_v = cp.take<double>((size_t)n * k); double* p_o = cp.take<double>(ng * k);
  memcpy(p_v, values, vb);
  hipError_t e = hipMemcpyAsync(d_v, p_v, vb, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = psm_launch_interp_to_grid(d_v, k, h->d_vtx_m2g, h->d_wts_m2g, h->d_src_of_cell, fill, d_o, (int64_t)ng, st);
  if (e == hipSuccess) e = hipMemcpyAsync(p_o, d_o, ob, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = wait_stream(st);
  if (e == hipSuccess) memcpy(grid_out, p_o, ob);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("mesh_to_grid: ") + hipGetErrorString(e));
  return PSM_OK;
}


int psm_poisson_features(psm_handle* h, const double* ux, const double* uy, const double* dux, const double* duy,
                         const double* sdfunct, int32_t ny, int32_t nx, const double* params, float* grid_out) {
  if (!h) return PSM_ERR_ARG;
  if (!ux || !uy || !dux || !duy || !sdfunct || !params || !grid_out) return fail(h, PSM_ERR_ARG, "null argument");
  if (ny < 2 || nx < 2 || (int64_t)ny * nx > ((int64_t)1 << 26)) return fail(h, PSM_ERR_ARG, "grid must be at least 2x2 (np.gradient)");
  if (!(params[1] != 0.0)) return fail(h, PSM_ERR_ARG, "U must be non-zero");
  for (int q = 3; q < 7; ++q)
    if (!(params[q] != 0.0)) return fail(h, PSM_ERR_ARG, "max_abs scales must be non-zero");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t n = (size_t)ny * nx, nwg = (n + 255) / 256;
  int rc;
  const size_t ib = 5 * n * sizeof(double), gb = 4 * n * sizeof(float);
  if ((rc = scratch_reserve(h, carve_size({ib, n * sizeof(double), 2 * nwg * sizeof(double), gb}), carve_size({ib, gb})))) return rc;
  Carver cd{(char*)h->scr_dev}, cp{(char*)h->scr_pin};
  double* d_in = cd.take<double>(5 * n); double* d_term = cd.take<double>(n); double* d_part = cd.take<double>(2 * nwg);
  float* d_grid = cd.take<float>(4 * n);
  double* p_in = cp.take<double>(5 * n); float* p_grid = cp.take<float>(4 * n);
  const double* src[5] = {ux, uy, dux, duy, sdfunct};
  for (int q = 0; q < 5; ++q) memcpy(p_in + q * n, src[q], n * sizeof(double));
  hipError_t e = hipMemcpyAsync(d_in, p_in, ib, hipMemcpyHostToDevice, st);
  PsmFeatureArgs fa{};
  fa.ux = d_in; fa.uy = d_in + n; fa.dux = d_in + 2 * n; fa.duy = d_in + 3 * n; fa.sdf = d_in + 4 * n;
  fa.term = d_term; fa.partial = d_part; fa.grid = d_grid; fa.ny = ny; fa.nx = nx;
  fa.L = params[0]; fa.U = params[1]; fa.k = params[2];
  for (int q = 0; q < 4; ++q) fa.max_abs[q] = params[3 + q];
  if (e == hipSuccess) e = psm_launch_poisson_features(fa, st);
  if (e == hipSuccess) e = hipMemcpyAsync(p_grid, d_grid, gb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = wait_stream(st);
  if (e == hipSuccess) memcpy(grid_out, p_grid, gb);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("poisson features: ") + hipGetErrorString(e));
  return PSM_OK;
}


int psm_set_integration(psm_handle* h, int32_t ny, int32_t nx, const double* sdfunct, int32_t cy, int32_t cx, double dx, double dy) {
  if (!h) return PSM_ERR_ARG;
  if (!sdfunct || ny < 2 || nx < 3) return fail(h, PSM_ERR_ARG, "bad integration geometry");
  if (cy < 1 || cy >= ny || cx < 1 || cx >= nx) return fail(h, PSM_ERR_ARG, "cut outside the grid");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const int wl = cx, wr = nx - cx + 1, hmax = std::max(cy, ny - cy);
  // "reset" quirk (Eval_dual_Dense_onlycil.py:394-396): nn = sdfunct[i,:].astype(int) indexes the block row
  std::vector<int2> fix((size_t)hmax * PSM_INTEG_MAX_FIX, make_int2(-1, -1));
  for (int a = 0; a < hmax; ++a) {
    std::map<int, int> last;                       // index value -> last position
    std::vector<int> nn(nx);
    for (int k = 0; k < nx; ++k) {
      nn[k] = (int)sdfunct[(int64_t)a * nx + k];   // C truncation == astype(int) for finite values
      if (nn[k] < 0) nn[k] += std::min(wl, wr);    // negative indices wrap in NumPy; not expected for a distance
      last[nn[k]] = k;
    }
    if ((int)last.size() > PSM_INTEG_MAX_FIX) return fail(h, PSM_ERR_UNSUPPORTED, "more distinct int(sdf) values on a row than supported");
    int e = 0;
    for (auto& kv : last) {
      if (kv.first >= std::min(wl, wr)) return fail(h, PSM_ERR_UNSUPPORTED, "int(sdfunct) indexes outside a quadrant row (the reference raises IndexError)");
      fix[(size_t)a * PSM_INTEG_MAX_FIX + e++] = make_int2(kv.first, kv.second > 0 ? nn[kv.second - 1] : -1);
    }
  }
  std::vector<int2> pairs;
  int npair[2];
  for (int q = 0; q < 2; ++q) {
    const int r0 = q ? cy : 0, r1 = q ? ny : cy;
    std::vector<int> rl, rr;
    for (int y = r0; y < r1; ++y) {
      if (sdfunct[(int64_t)y * nx + cx] != 0.0) rl.push_back(y);        // mask2 / mask4 (column cx)
      if (sdfunct[(int64_t)y * nx + cx - 1] != 0.0) rr.push_back(y);    // mask1 / mask3 (column cx-1)
    }
    if (rl.size() != rr.size()) return fail(h, PSM_ERR_UNSUPPORTED, "flow-cell counts of the two cut columns differ (the reference raises a broadcast error)");
    npair[q] = (int)rl.size();
    for (size_t k = 0; k < rl.size(); ++k) pairs.push_back(make_int2(rl[k], rr[k]));
  }
  int rc;
  if ((rc = dev_upload(h, &h->d_fixups, fix))) return rc;
  if (pairs.empty()) pairs.push_back(make_int2(0, 0));
  if ((rc = dev_upload(h, &h->d_pairs, pairs))) return rc;
  const size_t nbuf = (size_t)ny * wl + (size_t)ny * wr + 2 * (size_t)ny + 2 + (size_t)ny * nx;
  if ((rc = dev_alloc(h, &h->d_integ_buf, nbuf))) return rc;
  if ((rc = dev_alloc(h, &h->d_gradp, (size_t)ny * nx * 2))) return rc;
  PsmIntegArgs& a = h->integ;
  a.gradp = h->d_gradp; a.fixups = h->d_fixups; a.pairs = h->d_pairs; a.npair[0] = npair[0]; a.npair[1] = npair[1];
  a.rxl = h->d_integ_buf; a.rxr = a.rxl + (size_t)ny * wl; a.yl = a.rxr + (size_t)ny * wr; a.yr = a.yl + ny;
  a.corr = a.yr + ny; a.p_out = a.corr + 2;
  a.ny = ny; a.nx = nx; a.cy = cy; a.cx = cx; a.dx = (float)dx; a.dy = (float)dy;
  h->have_integ = true;
  return PSM_OK;
}


int psm_integrate_gradp(psm_handle* h, const float* gradp, float* p_out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->have_integ) return fail(h, PSM_ERR_STATE, "psm_set_integration has not been called");
  if (!gradp || !p_out) return fail(h, PSM_ERR_ARG, "null buffer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  const size_t n = (size_t)h->integ.ny * h->integ.nx;
  int rc;
  if ((rc = scratch_reserve(h, 0, carve_size({n * 2 * sizeof(float), n * sizeof(float)})))) return rc;
  Carver cp{(char*)h->scr_pin};
  float* p_g = cp.take<float>(n * 2); float* p_p = cp.take<float>(n);
  memcpy(p_g, gradp, n * 2 * sizeof(float));
  HIPCHK(h, hipMemcpyAsync(h->d_gradp, p_g, n * 2 * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, psm_launch_integrate(h->integ, st));
  HIPCHK(h, hipMemcpyAsync(p_p, h->integ.p_out, n * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(h, wait_stream(st));
  memcpy(p_out, p_p, n * sizeof(float));
  return PSM_OK;
}

}  // extern "C"
