// psm_api_plan.cpp -- C-ABI of libpsm_hip.so (include/psm.h): grid plan and bound geometries.  See psm_handle.h for the map of the five files.
#include "psm_handle.h"

namespace psm_impl {


// flags of one solve's guard waves: zero until a wave finds a mismatch (every wave rewrites its flag on every solve)
int ws_alloc_guard(psm_handle* h, Workspace& w) {
  if (!h->guard_waves) return PSM_OK;
  int rc = dev_alloc(h, &w.d_gflags, (size_t)h->guard_waves);
  if (rc) return rc;
  HIPCHK(h, hipMemset(w.d_gflags, 0, (size_t)h->guard_waves * sizeof(float)));
  return PSM_OK;
}


int ws_alloc(psm_handle* h, Workspace& w) {
  int rc;
  if ((rc = dev_alloc(h, &w.d_part, (size_t)h->n_slices * h->Mpad_cap * h->ld_in))) return rc;
  if ((rc = dev_alloc(h, &w.d_xin, (size_t)h->Mpad_cap * h->ld_in))) return rc;
  if ((rc = dev_alloc(h, &w.d_act[0], (size_t)h->Mpad_cap * h->max_width))) return rc;
  if ((rc = dev_alloc(h, &w.d_act[1], (size_t)h->Mpad_cap * h->max_width))) return rc;
  if (h->Mpad_cap > 128 && (rc = dev_alloc(h, &w.d_act_rows, (size_t)h->Mpad_cap * h->max_width))) return rc;   // only batches beyond 128 block rows pack
  if ((rc = dev_alloc(h, &w.d_res, (size_t)h->Mpad_cap * h->ld_out))) return rc;
  if ((rc = dev_alloc(h, &w.d_pred, (size_t)h->Mcap * h->K_out))) return rc;
  if ((rc = dev_alloc(h, &w.d_row_scale, (size_t)h->Mpad_cap))) return rc;
  if ((rc = dev_alloc(h, &w.d_spart, (size_t)h->cfg.max_cases * h->B * h->n_bands * h->plan.cp.NS))) return rc;
  if (h->cfg.variant == PSM_VARIANT_GRADP)
    if ((rc = dev_alloc(h, &w.d_colpart, (size_t)h->cfg.max_cases * h->n_bands * 128))) return rc;
  if ((rc = dev_alloc(h, &w.d_offs, (size_t)h->cfg.max_cases * h->cfg.c_out * h->B))) return rc;
  if ((rc = dev_alloc(h, &w.d_shift, (size_t)h->cfg.max_cases * h->cfg.c_out))) return rc;
  // padding rows / columns of the slabs and activations are read by the kernels: they must stay zero
  HIPCHK(h, hipMemset(w.d_part, 0, (size_t)h->n_slices * h->Mpad_cap * h->ld_in * sizeof(float)));
  HIPCHK(h, hipMemset(w.d_act[0], 0, (size_t)h->Mpad_cap * h->max_width * sizeof(float)));
  HIPCHK(h, hipMemset(w.d_act[1], 0, (size_t)h->Mpad_cap * h->max_width * sizeof(float)));
  if (w.d_act_rows) HIPCHK(h, hipMemset(w.d_act_rows, 0, (size_t)h->Mpad_cap * h->max_width * sizeof(float)));
  if (h->bound && h->bound_dots) { if ((rc = dev_alloc(h, &w.d_dots, h->bound_dots))) return rc; }
  if (h->bound && (rc = ws_alloc_guard(h, w))) return rc;
  if (h->bound && h->bound_cf) { if ((rc = dev_alloc(h, &w.d_dots2, h->cf_rows_all))) return rc; }
  for (int q = 0; q < 2 && h->c1_stride; ++q) {             // padding columns of the last layer's rows are read by the dense kernel
    if ((rc = dev_alloc(h, &w.d_c1[q], (size_t)h->Mpad_cap * h->c1_stride))) return rc;
    HIPCHK(h, hipMemset(w.d_c1[q], 0, (size_t)h->Mpad_cap * h->c1_stride * sizeof(float)));
  }
  return PSM_OK;
}


// Closed form of the offset chain for a bound case batch.  On a bound geometry every branch of the chain (np.isnan tests,
// the 0.9 coverage test, the first non-empty column) is decided by the strip COUNTS, so the correction of block b plus the
// global shift is a fixed linear map of the strip means: read off the host replay of the chain (psm_chain, double) by
// probing it with unit vectors, checked against a random probe, and folded into one table row per (field, block, source
// block) -- a linear combination of the strip / shift rows the bind kernels have just built.  Leaves bound_cf false (the
// chain launch stays) if the probe disagrees.
int build_closed_form(psm_handle* h, int n_cases, int rows, int Kh) {
  const int C = h->cfg.c_out, B = h->B, nst = h->n_strips, NS = h->plan.cp.NS;
  std::vector<float> hcnt((size_t)rows * n_cases);
  HIPCHK(h, psm_copy_d2h(hcnt.data(), h->d_cnt, hcnt.size() * sizeof(float)));
  const double qnan = std::nan("");
  std::vector<int32_t> ptr(1, 0), src, row_of_p;
  std::vector<float> coef, a0((size_t)n_cases * C * B);
  std::vector<std::vector<int>> strips_of(B);
  for (int s = 0; s < nst; ++s) strips_of[h->plan.strips[s].data].push_back(s);
  std::vector<double> mean(nst), cnt(nst), up(PSM_MAX_COLS), offs0(B), offs1(B), L((size_t)B * nst), A((size_t)B * nst);
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / (double)(1ull << 53) - 0.5; };
  for (int cs = 0; cs < n_cases; ++cs)
    for (int f = 0; f < C; ++f) {
      for (int s = 0; s < nst; ++s) cnt[s] = hcnt[(size_t)cs * rows + (size_t)f * nst + s];
      auto run = [&](std::vector<double>& out) {
        std::fill(up.begin(), up.end(), 0.0);
        PsmArrayChainCtx<double> cx{h->plan.blocks.data(), mean.data(), cnt.data(), NS, h->plan.cp.col_base, h->S, up.data(), out.data()};
        psm_chain<double>(h->plan.cp, cx, f);
      };
      for (int s = 0; s < nst; ++s) mean[s] = cnt[s] > 0 ? 0.0 : qnan;
      run(offs0);
      std::fill(L.begin(), L.end(), 0.0);
      for (int s = 0; s < nst; ++s) {
        if (!(cnt[s] > 0)) continue;
        mean[s] = 1.0;
        run(offs1);
        mean[s] = 0.0;
        for (int b = 0; b < B; ++b) { const double v = offs1[b] - offs0[b]; L[(size_t)b * nst + s] = (v == v) ? v : 0.0; }
      }
      // sub[b] = offs[b] + shift, shift = (sum of the shift partials) / (3 L_f) - sum_{w != 0} w[b'] offs[b']
      const float* w = h->h_shiftW.data() + (size_t)f * B;
      double base_shift = 0.0;
      std::vector<double> sh(nst, 0.0);
      for (int b2 = 0; b2 < B; ++b2) {
        if (w[b2] == 0.f) continue;
        base_shift += (double)w[b2] * offs0[b2];
        for (int s = 0; s < nst; ++s) sh[s] += (double)w[b2] * L[(size_t)b2 * nst + s];
      }
      for (int b = 0; b < B; ++b) {
        a0[((size_t)cs * C + f) * B + b] = (float)(offs0[b] - base_shift);       // NaN for skipped blocks / a poisoned shift, like the chain
        for (int s = 0; s < nst; ++s) A[(size_t)b * nst + s] = L[(size_t)b * nst + s] - sh[s];
      }
      // random probe: the chain itself against base + A . means
      for (int s = 0; s < nst; ++s) mean[s] = cnt[s] > 0 ? rnd() : qnan;
      run(offs1);
      double tsh = 0.0;
      for (int b2 = 0; b2 < B; ++b2) if (w[b2] != 0.f) tsh += (double)w[b2] * offs1[b2];
      for (int b = 0; b < B; ++b) {
        double want = offs1[b] - tsh, got = (double)a0[((size_t)cs * C + f) * B + b];
        for (int s = 0; s < nst; ++s) if (A[(size_t)b * nst + s] != 0.0) got += A[(size_t)b * nst + s] * mean[s];
        const bool wn = want != want, gn = got != got;
        if (wn != gn || (!wn && std::fabs(want - got) > 1e-5 * (1.0 + std::fabs(want)))) return PSM_OK;   // not linear: keep the chain launch
      }
      const double inv3L = h->plan.shiftA[f].empty() ? 0.0 : 1.0 / (3.0 * (double)h->plan.shiftA[f].size());
      for (int b = 0; b < B; ++b)
        for (int blk = 0; blk < B; ++blk) {
          for (int s : strips_of[blk]) {
            const double a = A[(size_t)b * nst + s];
            if (a != 0.0 && cnt[s] > 0) { src.push_back(f * nst + s); coef.push_back((float)(a / cnt[s])); }
          }
          src.push_back(C * nst + f * B + blk); coef.push_back((float)inv3L);          // the block's share of the shift's gathered part
          ptr.push_back((int32_t)src.size());
          row_of_p.push_back(cs * B + blk);
        }
    }
  const size_t pairs_pc = (size_t)C * B * B, pairs = pairs_pc * n_cases;
  int rc;
  int32_t *d_ptr = nullptr, *d_src = nullptr;
  float* d_coef = nullptr;
  if ((rc = dev_upload(h, &d_ptr, ptr)) || (rc = dev_upload(h, &d_src, src)) || (rc = dev_upload(h, &d_coef, coef))) return rc;
  std::vector<float> ones(pairs, 1.f);
  if ((rc = dev_alloc(h, &h->d_g2p, pairs * Kh)) || (rc = dev_alloc(h, &h->d_c2p, pairs)) || (rc = dev_upload(h, &h->d_cntp, ones)) ||
      (rc = dev_upload(h, &h->d_cfa0, a0)) || (rc = dev_upload(h, &h->d_row_of_p, row_of_p))) { dev_free(d_ptr); dev_free(d_src); dev_free(d_coef); return rc; }
  hipError_t e = hipSuccess;
  for (int cs = 0; cs < n_cases && e == hipSuccess; ++cs) {
    PsmPairFoldArgs pa{d_ptr + (size_t)cs * pairs_pc, d_src, d_coef, h->d_g2 + (size_t)cs * rows * Kh, h->d_c2 + (size_t)cs * rows,
                       h->d_g2p + (size_t)cs * pairs_pc * Kh, h->d_c2p + (size_t)cs * pairs_pc, (int)pairs_pc, Kh};
    e = psm_launch_pair_fold(pa, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  dev_free(d_ptr); dev_free(d_src); dev_free(d_coef);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("psm_launch_pair_fold: ") + hipGetErrorString(e));
  h->cf_rows_all = pairs;
  h->bound_cf = true;
  return PSM_OK;
}


// Bind the geometry (the flow-cell masks) of the planned grid: builds the tables of the 6-launch solve.
int bind_geometry_device(psm_handle* h, const float* d_grid, int n_cases) {
  const int nl = (int)h->dense.size();
  h->bound = false;
  const bool bf16 = h->cfg.precision == PSM_PRECISION_BF16;
  // the chain runs row-parallel in one wave (lane = block column); more than 64 blocks take the two-launch form of the
  // case batches (chain launch + chunked decode + paste), f32 only
  const bool small = h->B <= 64;
  if (h->plan.cp.n_x >= 64 || h->B > 4096 || h->ld_out > 128 || nl < 2 || !h->d_comp_nat || getenv("PSM_NO_FUSED_ASSEMBLE") != nullptr)
    return fail(h, PSM_ERR_UNSUPPORTED, "geometry binding needs < 64 block columns, <= 128 output components and a hidden layer");
  // bf16: the decode rounds `res`, so the head layer cannot be folded into the tables: rows over the ld_out components
  const int Kh = bf16 ? h->ld_out : h->dense[nl - 1].Kpad, C = h->cfg.c_out;
  if (Kh % 4 != 0 || Kh > 1024 || (small && C * h->n_strips + h->n_strips > 2560) || (size_t)(C * h->n_strips + h->n_strips + C * h->B) * 4 > 60 * 1024)
    return fail(h, PSM_ERR_UNSUPPORTED, "geometry binding: last hidden layer wider than 1024 or too many strips");
  if (n_cases > 1 && round_up(n_cases * h->B, 32) > 128 * 64) return fail(h, PSM_ERR_UNSUPPORTED, "geometry binding: too many block rows");
  const int rows = C * h->n_strips + C * h->B;
  const size_t all = (size_t)rows * n_cases;
  destroy_graphs(h);
  int rc;
  double *d_G = nullptr, *d_M = nullptr;
  if ((rc = dev_alloc(h, &d_G, (size_t)rows * h->ld_out))) return rc;
  if ((rc = dev_alloc(h, &d_M, (size_t)rows))) { dev_free(d_G); return rc; }
  if ((rc = dev_alloc(h, &h->d_g2, all * Kh)) || (rc = dev_alloc(h, &h->d_c2, all)) || (rc = dev_alloc(h, &h->d_cnt, all)) ||
      (rc = dev_alloc(h, &h->ws0.d_dots, all)) || (rc = dev_alloc(h, &h->d_row_of, all)) ||
      (rc = dev_alloc(h, &h->d_ownbits, (size_t)n_cases * h->B * (h->S * h->S / 32)))) { dev_free(d_G); dev_free(d_M); return rc; }
  const DenseLayer& hd = h->dense[nl - 1];
  hipError_t e = hipSuccess;
  for (int cs = 0; cs < n_cases && e == hipSuccess; ++cs) {
    PsmBindArgs a{};
    a.grid = d_grid + (size_t)cs * h->Ny * h->Nx * h->cfg.c_in;
    a.strips = h->d_strips; a.blk_y0x0 = h->d_blk; a.comp = h->d_comp_nat; a.mean = h->d_mean_out; a.owner = h->d_owner;
    a.shiftOwnA = h->d_shiftOwnA; a.shiftOwnB = h->d_shiftOwnB; a.Lmax = h->Lmax;
    for (int f = 0; f < 2; ++f) a.shiftL[f] = (int)h->plan.shiftA[f].size();
    a.Wh = hd.W; a.ldw = hd.ldw; a.Kh = Kh; a.bh = hd.b; a.sa = h->d_sa; a.sb = h->d_sb;
    a.G = d_G; a.Mrow = d_M;
    a.g2 = h->d_g2 + (size_t)cs * rows * Kh; a.c2 = h->d_c2 + (size_t)cs * rows; a.cnt = h->d_cnt + (size_t)cs * rows;
    a.row_of = h->d_row_of + (size_t)cs * rows; a.ownbits = h->d_ownbits + (size_t)cs * h->B * (h->S * h->S / 32);
    a.nst = h->n_strips; a.B = h->B; a.S = h->S; a.c_in = h->cfg.c_in; a.c_out = C; a.sdf_ch = h->cfg.sdf_channel;
    a.Ny = h->Ny; a.Nx = h->Nx; a.ld_out = h->ld_out; a.row_base = cs * h->B;
    e = bf16 ? psm_launch_bind_unfolded(a, h->stream) : psm_launch_bind(a, h->stream);   // same stream: the scratch is reused case after case
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  dev_free(d_G); dev_free(d_M);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("psm_launch_bind: ") + hipGetErrorString(e));
  h->bound_zero_fill = false;
  for (int32_t o : h->plan.owner) if (o < 0) { h->bound_zero_fill = true; break; }
  {                                                           // host copy of the bound flow-cell pattern (contract checks)
    const size_t npix = (size_t)h->Ny * h->Nx, cin = h->cfg.c_in;
    std::vector<float> g((size_t)n_cases * npix * cin);
    HIPCHK(h, psm_copy_d2h(g.data(), d_grid, g.size() * sizeof(float)));
    h->bound_mask.resize((size_t)n_cases * npix);
    for (size_t q = 0; q < (size_t)n_cases * npix; ++q) h->bound_mask[q] = g[q * cin + h->cfg.sdf_channel] != 0.f ? 1 : 0;
    // the same pattern as the guard waves see it: one 64-pixel ballot per word (pixels beyond the end clamp to the last)
    const size_t T = (size_t)n_cases * npix;
    h->guard_ballots = (int)((T + 63) / 64);
    h->guard_waves = (h->guard_ballots + PSM_GUARD_BALLOTS - 1) / PSM_GUARD_BALLOTS;
    std::vector<unsigned long long> bits((size_t)h->guard_ballots, 0ull);
    for (size_t w = 0; w < bits.size(); ++w)
      for (int l = 0; l < 64; ++l)
        if (h->bound_mask[std::min(w * 64 + l, T - 1)]) bits[w] |= 1ull << l;
    if ((rc = dev_upload(h, &h->d_maskbits, bits))) return rc;
    if ((rc = ws_alloc_guard(h, h->ws0))) return rc;
  }
  h->bound_rows = rows;
  h->bound_cases = n_cases;
  h->bound_dots = all;
  h->bound_cf = false;
  // The closed form trades the chain for a longer dots table: C B^2 rows per case against C (strips + B).  It is used where
  // that table stays small next to what the head launch streams anyway (deltas / chapter5 batches: fewer rows than the
  // strips; a single U_to_gradP case: 3.7 MB); a batch of U_to_gradP cases (1800 rows per case against 728) keeps the
  // chain launch.
  const double cf_bytes = (double)n_cases * C * h->B * h->B * Kh * 4.0, strip_bytes = (double)all * Kh * 4.0;
  if (h->B <= 64 && getenv("PSM_NO_CLOSED_FORM") == nullptr && cf_bytes <= std::max(8.0e6, 1.5 * strip_bytes)) {
    if ((rc = build_closed_form(h, n_cases, rows, Kh))) return rc;
    if (h->bound_cf && (rc = dev_alloc(h, &h->ws0.d_dots2, h->cf_rows_all))) return rc;
  }
  h->bound = true;
  if (h->ring_ready)
    for (auto& s : h->slot) {
      if ((rc = dev_alloc(h, &s.ws.d_dots, all)) || (rc = ws_alloc_guard(h, s.ws)) ||
          (h->bound_cf && (rc = dev_alloc(h, &s.ws.d_dots2, h->cf_rows_all)))) { h->bound = false; return rc; }
    }
  return PSM_OK;
}

}  // namespace psm_impl

// ============================================================================
extern "C" {


int psm_plan_grid(psm_handle* h, int32_t ny, int32_t nx) {
  if (!h) return PSM_ERR_ARG;
  if (!model_complete(h)) return fail(h, PSM_ERR_STATE, "model incomplete: call psm_set_pca, psm_set_scaler and psm_set_dense for every layer first");
  for (size_t l = 1; l < h->dense.size(); ++l)
    if (h->dense[l - 1].n_out != h->dense[l].n_in) return fail(h, PSM_ERR_ARG, "dense layers do not chain");
  for (const DenseLayer& d : h->dense)
    if (d.ln && d.ln_residual && d.n_in != d.n_out) return fail(h, PSM_ERR_ARG, "a LayerNormalization with the residual x + input needs a square layer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->bound = false;
  free_plan(h);
  std::string err;
  int rc = psm_build_plan(h->cfg.variant, ny, nx, h->S, h->ov, h->cfg.strict_degenerate != 0, h->plan, err);
  if (rc) return fail(h, rc, err);
  h->Ny = ny; h->Nx = nx; h->B = (int)h->plan.blocks.size();
  h->Mcap = h->cfg.max_cases * h->B;
  h->Mpad_cap = round_up(h->Mcap, 32);
  h->n_strips = (int)h->plan.strips.size();
  h->max_width = h->ld_in;
  for (auto& d : h->dense) h->max_width = std::max(h->max_width, d.ldw);
  h->c1_stride = 0;
  for (size_t q = 0; q < h->conv1d.size(); ++q) {
    if (q > 0 && h->conv1d[q - 1].cout != h->conv1d[q].cin) return fail(h, PSM_ERR_ARG, "Conv1D layers do not chain");
    h->c1_stride = std::max<int64_t>(h->c1_stride, round_up(h->cfg.p_in * h->conv1d[q].cout, 32));
  }
  if (!h->conv1d.empty() && h->dense[0].n_in != h->cfg.p_in * h->conv1d.back().cout)
    return fail(h, PSM_ERR_ARG, "first Dense layer after the Conv1D stack must take p_in * filters inputs (call psm_set_conv1d before psm_set_dense)");
  const size_t npix = (size_t)ny * nx;
  h->n_bands = h->S / PSM_STRIP_BAND;
  if ((rc = ws_alloc(h, h->ws0))) return rc;
  if ((rc = dev_alloc(h, &h->d_stamps, (size_t)16))) return rc;
  HIPCHK(h, hipMemset(h->d_stamps, 0, 16 * sizeof(unsigned long long)));
  if ((rc = dev_alloc(h, &h->d_grid_stage, (size_t)h->cfg.max_cases * npix * h->cfg.c_in))) return rc;
  if ((rc = dev_alloc(h, &h->d_fields_stage, (size_t)h->cfg.max_cases * npix * h->cfg.c_out))) return rc;
  HIPCHK(h, hipHostMalloc((void**)&h->h_grid, (size_t)h->cfg.max_cases * npix * h->cfg.c_in * sizeof(float), hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc((void**)&h->h_fields, (size_t)h->cfg.max_cases * npix * h->cfg.c_out * sizeof(float), hipHostMallocDefault));
  for (int i = 0; i < psm_handle::RING; ++i) {
    if (h->h_scale[i]) { (void)hipHostFree(h->h_scale[i]); h->h_scale[i] = nullptr; }
    HIPCHK(h, hipHostMalloc((void**)&h->h_scale[i], (size_t)h->Mpad_cap * sizeof(float), hipHostMallocDefault));
  }
  std::vector<float> ones(h->Mpad_cap, 1.f);
  if ((rc = dev_upload(h, &h->d_ones, ones))) return rc;
  std::vector<int64_t> rb(h->Mpad_cap, -1);
  for (int c = 0; c < h->cfg.max_cases; ++c)
    for (int b = 0; b < h->B; ++b)
      rb[(size_t)c * h->B + b] = (((int64_t)c * ny + h->plan.blocks[b].y0) * nx + h->plan.blocks[b].x0) * h->cfg.c_in;
  if ((rc = dev_upload(h, &h->d_row_base, rb))) return rc;
  std::vector<int32_t> st6((size_t)h->n_strips * 6), yx((size_t)h->B * 2);
  for (int e = 0; e < h->n_strips; ++e) {
    const PsmStrip& s = h->plan.strips[e];
    int32_t* o = &st6[(size_t)e * 6];
    o[0] = s.data; o[1] = s.mask; o[2] = s.r0; o[3] = s.r1; o[4] = s.c0; o[5] = s.c1;
  }
  for (int b = 0; b < h->B; ++b) { yx[2 * b] = h->plan.blocks[b].y0; yx[2 * b + 1] = h->plan.blocks[b].x0; }
  if ((rc = dev_upload(h, &h->d_strips, st6))) return rc;
  if ((rc = dev_upload(h, &h->d_blk, yx))) return rc;
  if ((rc = dev_upload(h, &h->d_blocks, h->plan.blocks))) return rc;
  if ((rc = dev_upload(h, &h->d_owner, h->plan.owner))) return rc;
  h->Lmax = (int)std::max(h->plan.shiftA[0].size(), h->plan.shiftA[1].size());
  std::vector<int32_t> sA((size_t)2 * h->Lmax, 0), sB((size_t)2 * h->Lmax, 0);
  for (int f = 0; f < 2; ++f) {
    std::copy(h->plan.shiftA[f].begin(), h->plan.shiftA[f].end(), sA.begin() + (size_t)f * h->Lmax);
    std::copy(h->plan.shiftB[f].begin(), h->plan.shiftB[f].end(), sB.begin() + (size_t)f * h->Lmax);
  }
  if ((rc = dev_upload(h, &h->d_shiftA, sA))) return rc;
  if ((rc = dev_upload(h, &h->d_shiftB, sB))) return rc;
  {
    std::vector<int32_t> oA(sA.size(), -1), oB(sB.size(), -1);
    std::vector<float> w((size_t)2 * h->B, 0.f);
    const int SS = h->S * h->S;
    for (int f = 0; f < 2; ++f) {
      const size_t L = h->plan.shiftA[f].size();
      std::vector<double> acc(h->B, 0.0);
      for (size_t k = 0; k < L; ++k) {
        const int a_ = h->plan.owner[h->plan.shiftA[f][k]], b_ = h->plan.owner[h->plan.shiftB[f][k]];
        oA[(size_t)f * h->Lmax + k] = a_; oB[(size_t)f * h->Lmax + k] = b_;
        if (a_ >= 0) acc[a_ / SS] += 3.0;
        if (b_ >= 0) acc[b_ / SS] -= 1.0;
      }
      for (int b = 0; b < h->B; ++b) w[(size_t)f * h->B + b] = L ? (float)(acc[b] / (3.0 * (double)L)) : 0.f;
    }
    if ((rc = dev_upload(h, &h->d_shiftOwnA, oA))) return rc;
    if ((rc = dev_upload(h, &h->d_shiftOwnB, oB))) return rc;
    if ((rc = dev_upload(h, &h->d_shiftW, w))) return rc;
    h->h_shiftW = w;
  }
  HIPCHK(h, hipDeviceSynchronize());
  {
    const char* ds = getenv("PSM_DEBUG_SKIP");
    h->debug_skip = ds ? atoi(ds) : 0;
  }
  {
    const char* nr = getenv("PSM_NO_FUSED_REDUCE");
    h->fuse_reduce_dense1 = !(nr && nr[0] == '1');
  }
  {
    const char* nf = getenv("PSM_NO_FUSED_ASSEMBLE");
    h->fused_assemble = (h->B <= 64 && h->plan.cp.n_x < 64) && !(nf && nf[0] == '1');
  }
  h->planned = true;
  return PSM_OK;
}


int psm_bind_geometry_cases(psm_handle* h, const float* grids, int32_t n_cases, int32_t on_device) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!grids) return fail(h, PSM_ERR_ARG, "null buffer");
  if (n_cases < 1 || n_cases > h->cfg.max_cases) return fail(h, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->bound_scope = 2;
  if (on_device) return bind_geometry_device(h, grids, n_cases);
  const size_t gin = (size_t)n_cases * h->Ny * h->Nx * h->cfg.c_in * sizeof(float);
  HIPCHK(h, psm_copy_h2d(h->d_grid_stage, grids, gin));
  return bind_geometry_device(h, h->d_grid_stage, n_cases);
}


int psm_bind_geometry(psm_handle* h, const float* grid, int32_t on_device) { return psm_bind_geometry_cases(h, grid, 1, on_device); }


int psm_unbind_geometry(psm_handle* h) {
  if (!h) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  destroy_graphs(h);
  h->bound = false;
  return PSM_OK;
}


int psm_geometry_bound(const psm_handle* h) { return (h && h->bound) ? 1 : 0; }


int psm_bound_mask(const psm_handle* h, uint8_t* mask, size_t cap) {
  if (!h || !mask) return PSM_ERR_ARG;
  if (!h->bound) return PSM_ERR_STATE;
  if (cap < h->bound_mask.size()) return PSM_ERR_ARG;
  memcpy(mask, h->bound_mask.data(), h->bound_mask.size());
  return PSM_OK;
}


int psm_num_blocks(const psm_handle* h) { return (h && h->planned) ? h->B : PSM_ERR_STATE; }


int psm_grid_shape(const psm_handle* h, int32_t* shape) {
  if (!h || !shape || !h->planned) return PSM_ERR_STATE;
  shape[0] = h->Ny; shape[1] = h->Nx; shape[2] = h->cfg.c_in; shape[3] = h->cfg.c_out;
  return PSM_OK;
}


// ---- host-only helpers -------------------------------------------------------
int psm_layout(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap, int32_t* blocks, int32_t cap,
               int32_t* n_x, int32_t* n_y) {
  std::vector<PsmBlock> b;
  std::string err;
  int nx_ = 0, ny_ = 0;
  int rc = psm_build_layout(variant, ny, nx, block, overlap, b, nx_, ny_, err);
  if (rc) return fail(nullptr, rc, err);
  if (n_x) *n_x = nx_;
  if (n_y) *n_y = ny_;
  if (blocks)
    for (int i = 0; i < (int)b.size() && i < cap; ++i) {
      blocks[4 * i] = b[i].y0; blocks[4 * i + 1] = b[i].x0; blocks[4 * i + 2] = b[i].ti; blocks[4 * i + 3] = b[i].tj;
    }
  return (int)b.size();
}


int psm_owner_map(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap, int32_t strict, int32_t* owner) {
  if (!owner) return fail(nullptr, PSM_ERR_ARG, "null owner buffer");
  PsmPlan plan;
  std::string err;
  int rc = psm_build_plan(variant, ny, nx, block, overlap, strict != 0, plan, err);
  if (rc) return fail(nullptr, rc, err);
  memcpy(owner, plan.owner.data(), plan.owner.size() * sizeof(int32_t));
  return PSM_OK;
}

}  // extern "C"
