// psm_api_introspect.cpp -- C-ABI of libpsm_hip.so (include/psm.h): stage read-back, profiling, kernel timing, host-side reference.  See psm_handle.h for the map of the five files.
#include "psm_handle.h"

namespace psm_impl {


// Dispatch-level time of EVERY kernel of the solve path: `steps` solves through the same launch sequence as
// psm_solve_grid_device, each dispatch stamped by hipExtLaunchKernelGGL (its own begin / end, what rocprofv3 reads).
// every dispatch of `steps` solves with its own begin / end stamps: per kernel (launch order of first appearance) the samples in ms
int collect_kernel_samples(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps,
                                  std::vector<std::string>& seen, std::vector<std::vector<float>>& samp) {
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (h->timed_kernel >= 0) return fail(h, PSM_ERR_STATE, "psm_enable_kernel_timing is active");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  PsmLaunchProbe probe;
  auto drain = [&]() -> int {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& r : probe.recs) {
      float t = 0.f;
      std::string nm(r.name);                       // "(psm_x_kernel<A, B>)" -> "psm_x_kernel<A, B>": the launcher's template
      while (!nm.empty() && (nm[0] == '(' || nm[0] == ' ')) nm.erase(0, 1);     // expression, distinct per instantiation family
      while (!nm.empty() && (nm.back() == ')' || nm.back() == ' ')) nm.pop_back();
      if (r.tag >= 0) {                             // the same instantiation serves several Dense layers: one entry per layer
        const std::string sfx = "#layer" + std::to_string(r.tag);
        if (nm.size() + sfx.size() > 63) nm.resize(63 - sfx.size());
        nm += sfx;
      }
      if (nm.size() > 63) nm.resize(63);
      if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
        size_t k = 0;
        while (k < seen.size() && seen[k] != nm) ++k;
        if (k == seen.size()) { seen.push_back(nm); samp.emplace_back(); }
        samp[k].push_back(t);
      }
      probe.pool.push_back(r.e0); probe.pool.push_back(r.e1);
    }
    probe.recs.clear();
    return PSM_OK;
  };
  const bool graph = h->use_graph;
  h->use_graph = false;                              // plain launches: every dispatch carries its own events
  int rc = PSM_OK;
  psm_launch_probe = &probe;
  for (int i = 0; i < steps && rc == PSM_OK; ++i) {
    rc = solve_device(h, d_grid, n_cases, nullptr, d_fields, h->stream, nullptr);
    if (rc == PSM_OK && (i % 64) == 63) rc = drain();
  }
  psm_launch_probe = nullptr;
  h->use_graph = graph;
  if (rc == PSM_OK) rc = drain(); else (void)hipStreamSynchronize(h->stream);
  for (auto& r : probe.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : probe.pool) (void)hipEventDestroy(e);
  return rc;
}

}  // namespace psm_impl

// ============================================================================
extern "C" {


int psm_read_stage(psm_handle* h, int32_t stage, float* dst, size_t dst_floats) {
  if (!h || !dst) return PSM_ERR_ARG;
  if (!h->planned || h->last_cases < 1) return fail(h, PSM_ERR_STATE, "no solve has run yet");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipDeviceSynchronize());
  const int M = h->last_cases * h->B;
  auto rows = [&](const float* src, int ld, int width) -> int {
    if (dst_floats < (size_t)M * width) return fail(h, PSM_ERR_ARG, "destination too small");
    HIPCHK(h, psm_copy_d2h_2d(dst, (size_t)width * sizeof(float), src, (size_t)ld * sizeof(float), (size_t)width * sizeof(float), M));
    return PSM_OK;
  };
  switch (stage) {
    case PSM_STAGE_X_INPUT: return rows(h->ws0.d_xin, h->ld_in, h->cfg.p_in);
    case PSM_STAGE_RES: return rows(h->ws0.d_res, h->ld_out, h->cfg.p_out);
    case PSM_STAGE_BLOCK_PRED: return rows(h->ws0.d_pred, h->K_out, h->K_out);
    case PSM_STAGE_OFFSETS:
    case PSM_STAGE_SHIFT:
      if (h->bound && h->bound_cf && h->last_used_cf && h->last_cases == h->bound_cases) {
        // the last solve took the closed form: run the chain itself once, from the strip means of the same activations
        const int nl = (int)h->dense.size();
        const bool bf = h->cfg.precision == PSM_PRECISION_BF16;
        const float* act = bf ? h->ws0.d_res : (h->last_act_packed ? h->ws0.d_act_rows : h->ws0.d_act[(nl - 2) & 1]);
        const int ld_act = bf ? h->ld_out : h->dense[nl - 2].ldw;
        PsmDotsArgs dd{h->d_g2, h->d_c2, h->d_cnt, h->d_row_of, h->last_row_scale ? h->last_row_scale : h->d_ones, h->ws0.d_dots,
                       h->bound_rows * h->last_cases, bf ? h->ld_out : h->dense[nl - 1].Kpad, PsmGuardArgs{}};
        HIPCHK(h, psm_launch_act_dots(dd, act, ld_act, bf ? 1 : 0, h->stream));
        PsmBoundBatchArgs bb{};
        bb.cp = h->plan.cp; bb.blocks = h->d_blocks; bb.dots = h->ws0.d_dots; bb.scnt = h->d_cnt; bb.ownbits = h->d_ownbits;
        bb.blk_y0x0 = h->d_blk; bb.shiftW = h->d_shiftW;
        for (int f = 0; f < 2; ++f) bb.shiftL[f] = (int)h->plan.shiftA[f].size();
        bb.offs = h->ws0.d_offs; bb.shift = h->ws0.d_shift; bb.Nx = h->Nx; bb.npix = h->Ny * h->Nx;
        bb.n_strips = h->n_strips; bb.B = h->B; bb.rows_pc = h->bound_rows; bb.n_cases = h->last_cases;
        bb.gflags = h->d_gzero; bb.n_gwaves = 1;
        HIPCHK(h, psm_launch_chain_dots(bb, h->cfg.c_out, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
      }
      if (stage == PSM_STAGE_SHIFT) {
        const size_t n = (size_t)h->last_cases * h->cfg.c_out;
        if (dst_floats < n) return fail(h, PSM_ERR_ARG, "destination too small");
        HIPCHK(h, psm_copy_d2h(dst, h->ws0.d_shift, n * sizeof(float)));
        return PSM_OK;
      }
      {
      const size_t n = (size_t)h->last_cases * h->cfg.c_out * h->B;
      if (dst_floats < n) return fail(h, PSM_ERR_ARG, "destination too small");
      HIPCHK(h, psm_copy_d2h(dst, h->ws0.d_offs, n * sizeof(float)));
      return PSM_OK;
    }
    case 6: {   // diagnostic builds only: raw stamps of workgroup 0, microseconds after the earliest one
      unsigned long long t[64];
      if (dst_floats < 64) return fail(h, PSM_ERR_ARG, "destination too small");
      HIPCHK(h, psm_read_stamps(t));
      unsigned long long t0 = ~0ull;
      for (int k = 0; k < 64; ++k) if (t[k] && t[k] < t0) t0 = t[k];
      for (int k = 0; k < 64; ++k) dst[k] = t[k] ? (float)((double)(t[k] - t0) * 0.01) : -1.f;
      return PSM_OK;
    }
    case 5: {   // diagnostic builds only: stamp deltas of workgroup 0 in microseconds
      unsigned long long t[64];
      if (dst_floats < 64) return fail(h, PSM_ERR_ARG, "destination too small");
      HIPCHK(h, psm_read_stamps(t));
      for (int k = 0; k < 63; ++k) dst[k] = (t[k + 1] && t[k]) ? (float)((double)t[k + 1] * 0.01 - (double)t[k] * 0.01) : 0.f;
      dst[63] = 0.f;
      return PSM_OK;
    }
  }
  return fail(h, PSM_ERR_ARG, "unknown stage");
}


int psm_profile_solve(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, float* ms) {
  if (!h || !ms) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipEvent_t ev[PSM_K_COUNT + 1];
  for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
  int rc = solve_device(h, d_grid, n_cases, nullptr, d_fields, h->stream, ev);
  if (rc == PSM_OK) {
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(h, PSM_ERR_HIP, hipGetErrorString(e));
  }
  if (rc == PSM_OK)
    for (int k = 0; k < PSM_K_COUNT; ++k) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ev[k], ev[k + 1]);
      ms[k] = t;
    }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}


int psm_enable_kernel_timing(psm_handle* h, int32_t kernel, int32_t on) {
  if (!h) return PSM_ERR_ARG;
  if (kernel < 0 || kernel >= PSM_K_COUNT) return fail(h, PSM_ERR_ARG, "unknown kernel group");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipDeviceSynchronize());
  for (auto& p : h->timed_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  h->timed_events.clear();
  h->timed_total_ms = 0.0; h->timed_launches = 0;
  h->timed_kernel = on ? kernel : -1;
  h->timed_repeat = on > 1 ? (on > 64 ? 64 : on) : 1;
  return PSM_OK;
}


int psm_get_kernel_timing(psm_handle* h, int32_t kernel, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return PSM_ERR_ARG;
  if (kernel != h->timed_kernel) return fail(h, PSM_ERR_STATE, "timing is not enabled for this kernel group");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipDeviceSynchronize());
  for (auto& p : h->timed_events) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) { h->timed_total_ms += t; h->timed_launches += h->timed_repeat; }
    (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
  }
  h->timed_events.clear();
  *total_ms = h->timed_total_ms; *launches = h->timed_launches;
  return PSM_OK;
}


int psm_time_kernels(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps, char* names,
                     double* total_ms, int64_t* launches, int32_t cap, int32_t* n_kernels) {
  if (!h || !names || !total_ms || !launches || !n_kernels || cap < 1 || steps < 1) return PSM_ERR_ARG;
  std::vector<std::string> seen;
  std::vector<std::vector<float>> samp;
  int rc = collect_kernel_samples(h, d_grid, n_cases, d_fields, steps, seen, samp);
  if (rc) return rc;
  *n_kernels = (int32_t)seen.size();
  for (int k = 0; k < (int)seen.size() && k < cap; ++k) {
    snprintf(names + (size_t)k * 64, 64, "%s", seen[k].c_str());
    double tot = 0.0;
    for (float t : samp[k]) tot += t;
    total_ms[k] = tot; launches[k] = (int64_t)samp[k].size();
  }
  return PSM_OK;
}


// the same pass, per kernel the MEDIAN and the 10th / 90th percentile of its dispatch durations (microseconds): one slow dispatch
// (a clock dip, a page fault) moves a mean of 20-200 samples, not these
int psm_time_kernels_q(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps, char* names,
                       double* median_us, double* p10_us, double* p90_us, int64_t* launches, int32_t cap, int32_t* n_kernels) {
  if (!h || !names || !median_us || !launches || !n_kernels || cap < 1 || steps < 1) return PSM_ERR_ARG;
  std::vector<std::string> seen;
  std::vector<std::vector<float>> samp;
  int rc = collect_kernel_samples(h, d_grid, n_cases, d_fields, steps, seen, samp);
  if (rc) return rc;
  *n_kernels = (int32_t)seen.size();
  for (int k = 0; k < (int)seen.size() && k < cap; ++k) {
    snprintf(names + (size_t)k * 64, 64, "%s", seen[k].c_str());
    std::vector<float>& v = samp[k];
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    auto q = [&](double f) { return n ? (double)v[std::min(n - 1, (size_t)(f * (double)(n - 1) + 0.5))] * 1e3 : 0.0; };
    median_us[k] = q(0.5);
    if (p10_us) p10_us[k] = q(0.1);
    if (p90_us) p90_us[k] = q(0.9);
    launches[k] = (int64_t)n;
  }
  return PSM_OK;
}


int psm_event_pair_overhead(psm_handle* h, int32_t n, double* median_ms) {
  if (!h || !median_ms || n < 1 || n > 10000) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
  for (int i = 0; i < n; ++i) {           // an empty event pair per "launch": what the timing itself costs
    HIPCHK(h, hipEventRecord(ev[2 * i], h->stream));
    HIPCHK(h, hipEventRecord(ev[2 * i + 1], h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<float> t(n, 0.f);
  for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  *median_ms = t[n / 2];
  return PSM_OK;
}



// Host replay of the device reassembly (strip table -> chain -> owner-map paste) on
// caller-supplied decoded blocks.  Verification helper for the plan tables and the
// chain logic only: nothing in psm_solve_* calls it.
int psm_debug_reassemble_host(int32_t variant, int32_t ny, int32_t nx, int32_t block, int32_t overlap, int32_t strict,
                              int32_t c_in, int32_t c_out, int32_t sdf_ch, const float* grid, const float* pred,
                              float* fields, float* offsets, float* shifts) {
  if (!grid || !pred || !fields) return fail(nullptr, PSM_ERR_ARG, "null buffer");
  PsmPlan plan;
  std::string err;
  int rc = psm_build_plan(variant, ny, nx, block, overlap, strict != 0, plan, err);
  if (rc) return fail(nullptr, rc, err);
  const int S = block, SS = S * S, B = plan.cp.B, NSTR = (int)plan.strips.size();
  std::vector<float> sum(NSTR), cnt(NSTR), offs(B), up(PSM_MAX_COLS);
  for (int f = 0; f < c_out; ++f) {
    for (int e = 0; e < NSTR; ++e) {
      const PsmStrip& st = plan.strips[e];
      float s = 0.f, c = 0.f;
      for (int r = st.r0; r < st.r1; ++r)
        for (int cc = st.c0; cc < st.c1; ++cc) {
          bool on = true;
          if (st.mask >= 0) {
            const PsmBlock& mb = plan.blocks[st.mask];
            on = grid[((size_t)(mb.y0 + r) * nx + mb.x0 + cc) * c_in + sdf_ch] != 0.f;
          }
          if (on) { s += pred[((size_t)st.data * SS + r * S + cc) * c_out + f]; c += 1.f; }
        }
      sum[e] = s / c; cnt[e] = c;     // mean; 0/0 -> NaN like np.mean([])
    }
    for (auto& u : up) u = 0.f;
    PsmArrayChainCtx<float> cx{plan.blocks.data(), sum.data(), cnt.data(), plan.cp.NS, plan.cp.col_base, S, up.data(), offs.data()};
    psm_chain<float>(plan.cp, cx, f);
    double acc = 0.0;
    const size_t L = plan.shiftA[f].size();
    for (size_t k = 0; k < L; ++k) {
      const int oa = plan.owner[plan.shiftA[f][k]], ob = plan.owner[plan.shiftB[f][k]];
      const float va = oa >= 0 ? pred[(size_t)oa * c_out + f] - offs[oa / SS] : 0.f;
      const float vb = ob >= 0 ? pred[(size_t)ob * c_out + f] - offs[ob / SS] : 0.f;
      acc += 3.0 * va - vb;
    }
    const float shift = (float)(acc / (double)L / 3.0);
    for (size_t pix = 0; pix < (size_t)ny * nx; ++pix) {
      const int o = plan.owner[pix];
      fields[pix * c_out + f] = o >= 0 ? pred[(size_t)o * c_out + f] - offs[o / SS] - shift : 0.f;
    }
    if (offsets) memcpy(offsets + (size_t)f * B, offs.data(), B * sizeof(float));
    if (shifts) shifts[f] = shift;
  }
  return PSM_OK;
}

}  // extern "C"
