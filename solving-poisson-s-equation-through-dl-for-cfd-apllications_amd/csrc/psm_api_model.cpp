// psm_api_model.cpp -- C-ABI of libpsm_hip.so (include/psm.h): handle lifetime and model artefacts.  See psm_handle.h for the map of the five files.
#include "psm_handle.h"

namespace psm_impl { thread_local std::string g_create_error; }

namespace psm_impl {


// The synchronous entries last ~100 us: they poll the stream / event instead of sleeping in hip*Synchronize (the
// wake-up of a blocked thread alone costs 10-20 us per call) -- but only for a bounded time (PSM_SPIN_US, default
// 300 us), after which the thread blocks, and never when this process shares its cores with more MPI / torchrun
// ranks than it has cores (a spinning rank would then steal the time of another rank's solver thread).
// PSM_SYNC_BLOCK=1 forces blocking waits, PSM_SYNC_BLOCK=0 forces the bounded spin.
int local_ranks_from_env() {
  for (const char* k : {"OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "PMI_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "LOCAL_WORLD_SIZE"}) {
    const char* v = getenv(k);
    if (v && atoi(v) > 0) return atoi(v);
  }
  return 1;
}

bool sync_blocks() {
  static const bool b = [] {
    const char* e = getenv("PSM_SYNC_BLOCK");
    if (e) return e[0] != '0';
    cpu_set_t set;
    int cores = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    if (cores <= 0) cores = (int)std::thread::hardware_concurrency();
    return local_ranks_from_env() > cores;       // oversubscribed: sleep instead of spinning
  }();
  return b;
}

hipError_t wait_stream(hipStream_t st) {
  return bounded_wait([&] { return hipStreamQuery(st); }, [&] { return hipStreamSynchronize(st); });
}

hipError_t wait_event(hipEvent_t ev) {
  return bounded_wait([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); });
}


int fail(psm_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}


int scratch_reserve(psm_handle* h, size_t dev_bytes, size_t pin_bytes) {
  if (dev_bytes > h->scr_dev_cap) {
    if (h->scr_dev) { (void)hipStreamSynchronize(h->stream); (void)psm_dev_free(h->scr_dev); h->scr_dev = nullptr; h->scr_dev_cap = 0; }
    const size_t cap = dev_bytes + dev_bytes / 2;
    hipError_t e = psm_dev_malloc(&h->scr_dev, cap);
    if (e != hipSuccess) return fail(h, PSM_ERR_NOMEM, std::string("psm_dev_malloc(scratch): ") + hipGetErrorString(e));
    h->scr_dev_cap = cap;
  }
  if (pin_bytes > h->scr_pin_cap) {
    if (h->scr_pin) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->scr_pin); h->scr_pin = nullptr; h->scr_pin_cap = 0; }
    const size_t cap = pin_bytes + pin_bytes / 2;
    hipError_t e = hipHostMalloc(&h->scr_pin, cap, hipHostMallocDefault);
    if (e != hipSuccess) return fail(h, PSM_ERR_NOMEM, std::string("hipHostMalloc(scratch): ") + hipGetErrorString(e));
    h->scr_pin_cap = cap;
  }
  return PSM_OK;
}

void destroy_graphs(psm_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  if (h->mesh_graph) { (void)hipGraphExecDestroy(h->mesh_graph); h->mesh_graph = nullptr; }
  ring_drop_graphs(h);
}


void ws_free(Workspace& w) {
  dev_free(w.d_part); dev_free(w.d_xin); dev_free(w.d_act[0]); dev_free(w.d_act[1]); dev_free(w.d_act_rows); w.d_act_rows = nullptr; dev_free(w.d_res); dev_free(w.d_pred);
  dev_free(w.d_row_scale); dev_free(w.d_spart); dev_free(w.d_colpart); dev_free(w.d_offs); dev_free(w.d_shift); dev_free(w.d_dots);
  dev_free(w.d_gflags); dev_free(w.d_c1[0]); dev_free(w.d_c1[1]); dev_free(w.d_dots2);
}


void ring_drop_graphs(psm_handle* h) {
  for (auto& s : h->slot) {
    if ((s.g_full || s.g_kern) && s.st) (void)hipStreamSynchronize(s.st);     // a replay of this slot may still be running
    if (s.g_full) { (void)hipGraphExecDestroy(s.g_full); s.g_full = nullptr; }
    if (s.g_kern) { (void)hipGraphExecDestroy(s.g_kern); s.g_kern = nullptr; }
    s.g_full_key = s.g_kern_key = -1;
  }
}


void free_plan(psm_handle* h) {
  ring_drop_graphs(h);
  for (auto& s : h->slot) {
    if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_rs) (void)hipHostFree(s.h_rs);
    if (s.d_in) (void)psm_dev_free(s.d_in);
    if (s.d_out) (void)psm_dev_free(s.d_out);
    if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    ws_free(s.ws);
    const int gidx = s.ws.gidx;
    s = psm_handle::Slot{};
    s.ws.gidx = gidx;
  }
  h->ring_ready = false;
  destroy_graphs(h);
  ws_free(h->ws0);
  dev_free(h->d_row_base); dev_free(h->d_ones); dev_free(h->d_strips);
  dev_free(h->d_blk); dev_free(h->d_owner); dev_free(h->d_shiftA); dev_free(h->d_shiftB); dev_free(h->d_shiftOwnA); dev_free(h->d_shiftOwnB); dev_free(h->d_shiftW); dev_free(h->d_blocks);
  dev_free(h->d_stamps); dev_free(h->d_grid_stage); dev_free(h->d_fields_stage);
  if (h->h_grid) { (void)hipHostFree(h->h_grid); h->h_grid = nullptr; }
  if (h->h_fields) { (void)hipHostFree(h->h_fields); h->h_fields = nullptr; }
  h->planned = false;
  // The mesh-side tables (psm_set_geometry) index THIS plan's grid and its staging buffers, which are gone now: a later
  // psm_solve / psm_mesh_to_grid must fail with PSM_ERR_STATE until psm_set_geometry runs again, not launch on null buffers.
  // (The registered host arrays stay registered; the graph that holds their addresses goes with the plan.)
  h->have_geometry = false;
  if (h->mesh_graph) { (void)hipGraphExecDestroy(h->mesh_graph); h->mesh_graph = nullptr; }
}


// ---- weight packing ----------------------------------------------------------
// comp_in [P][K] (sklearn components_) -> [slice][ntile][G][64 lanes] float4 so that one wave
// instruction of the encode kernel reads 1 KiB contiguous; element j of lane l in group g is
// comp[32*t + (l&31)][slice*KS + 8*g + 4*(l>>5) + j]  (zero for padded components).
std::vector<float4> pack_comp_in(const double* comp, int P, int K, int c_in, int S, int NT) {
  const int KS = PSM_PIX_PER_SLICE * c_in, G = KS / 8, n_slices = S * S / PSM_PIX_PER_SLICE;
  std::vector<float4> out((size_t)n_slices * NT * G * 64);
  for (int s = 0; s < n_slices; ++s)
    for (int t = 0; t < NT; ++t)
      for (int g = 0; g < G; ++g)
        for (int l = 0; l < 64; ++l) {
          const int p = 32 * t + (l & 31);
          const int64_t k = (int64_t)s * KS + 8 * g + 4 * (l >> 5);
          float v[4] = {0, 0, 0, 0};
          if (p < P) for (int j = 0; j < 4; ++j) v[j] = (float)comp[(int64_t)p * K + k + j];
          out[(((size_t)s * NT + t) * G + g) * 64 + l] = make_float4(v[0], v[1], v[2], v[3]);
        }
  return out;
}


// comp_out [P][K_out] -> [coltile][Gd][64] float4: element j of lane l in group g is
// comp[8*g + 4*(l>>5) + j][32*ct + (l&31)]  (zero for padded components).
std::vector<float4> pack_comp_out(const double* comp, int P, int K_out, int Gd) {
  const int nct = K_out / 32;
  std::vector<float4> out((size_t)nct * Gd * 64);
  for (int ct = 0; ct < nct; ++ct)
    for (int g = 0; g < Gd; ++g)
      for (int l = 0; l < 64; ++l) {
        const int col = 32 * ct + (l & 31);
        float v[4];
        for (int j = 0; j < 4; ++j) {
          const int p = 8 * g + 4 * (l >> 5) + j;
          v[j] = p < P ? (float)comp[(int64_t)p * K_out + col] : 0.f;
        }
        out[((size_t)ct * Gd + g) * 64 + l] = make_float4(v[0], v[1], v[2], v[3]);
      }
  return out;
}


void unpin_buffers(psm_handle* h) {
  if (h->mesh_graph) { (void)hipGraphExecDestroy(h->mesh_graph); h->mesh_graph = nullptr; }      // it holds the registered addresses
  h->pinned_cells_dev = nullptr;
  if (h->pinned_cells) { (void)hipHostUnregister((void*)h->pinned_cells); h->pinned_cells = nullptr; }
  if (h->pinned_p) { (void)hipHostUnregister((void*)h->pinned_p); h->pinned_p = nullptr; h->pinned_p_dev = nullptr; }
}


void free_geometry(psm_handle* h) {
  unpin_buffers(h);
  dev_free(h->d_vtx_m2g); dev_free(h->d_src_of_cell); dev_free(h->d_vtx_g2m); dev_free(h->d_cell_of_point);
  dev_free(h->d_wts_m2g); dev_free(h->d_sdf); dev_free(h->d_wts_g2m); dev_free(h->d_cells); dev_free(h->d_p);
  dev_free(h->d_umax); dev_free(h->d_umax_part); dev_free(h->d_near_wall);
  dev_free(h->d_fixups); dev_free(h->d_pairs); dev_free(h->d_integ_buf); dev_free(h->d_gradp);
  h->have_integ = false;
  if (h->h_cells) { (void)hipHostFree(h->h_cells); h->h_cells = nullptr; }
  if (h->h_p) { (void)hipHostFree(h->h_p); h->h_p = nullptr; }
  h->have_geometry = false;
}


// bf16 tilings for v_mfma_f32_32x32x16_bf16: 16 bytes (8 bf16) per lane and MFMA step.
//  comp_in : [slice][ntile][g = KS/16][64]; element j of lane l = comp[32t+(l&31)][slice*KS + 16g + 8(l>>5) + j]
//  comp_out: [coltile][g = ld_out/16][64]; element j of lane l = comp[16g + 8(l>>5) + j][32ct + (l&31)]
std::vector<uint16_t> pack_comp_in_bf16(const double* comp, int P, int K, int c_in, int S, int NT) {
  const int KS = PSM_PIX_PER_SLICE * c_in, G = KS / 16, n_slices = S * S / PSM_PIX_PER_SLICE;
  std::vector<uint16_t> out((size_t)n_slices * NT * G * 64 * 8, 0);
  for (int s = 0; s < n_slices; ++s)
    for (int t = 0; t < NT; ++t)
      for (int g = 0; g < G; ++g)
        for (int l = 0; l < 64; ++l) {
          const int p = 32 * t + (l & 31);
          if (p >= P) continue;
          const int64_t k = (int64_t)s * KS + 16 * g + 8 * (l >> 5);
          uint16_t* o = &out[((((size_t)s * NT + t) * G + g) * 64 + l) * 8];
          for (int j = 0; j < 8; ++j) o[j] = f2bf(comp[(int64_t)p * K + k + j]);
        }
  return out;
}


std::vector<uint16_t> pack_comp_out_bf16(const double* comp, int P, int K_out, int G) {
  const int nct = K_out / 32;
  std::vector<uint16_t> out((size_t)nct * G * 64 * 8, 0);
  for (int ct = 0; ct < nct; ++ct)
    for (int g = 0; g < G; ++g)
      for (int l = 0; l < 64; ++l) {
        const int col = 32 * ct + (l & 31);
        uint16_t* o = &out[(((size_t)ct * G + g) * 64 + l) * 8];
        for (int j = 0; j < 8; ++j) {
          const int p = 16 * g + 8 * (l >> 5) + j;
          o[j] = p < P ? f2bf(comp[(int64_t)p * K_out + col]) : 0;
        }
      }
  return out;
}


bool model_complete(const psm_handle* h) {
  if (!h->have_pca || !h->have_scaler) return false;
  for (auto& d : h->dense) if (!d.set) return false;
  for (auto& c : h->conv1d) if (!c.set) return false;
  return true;
}

}  // namespace psm_impl

// ============================================================================
extern "C" {


int psm_abi_version(void) { return PSM_ABI_VERSION; }


const char* psm_last_error(const psm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }


int psm_create(const psm_config* cfg, psm_handle** out) {
  if (!cfg || !out) return fail(nullptr, PSM_ERR_ARG, "null argument");
  *out = nullptr;
  if (cfg->abi_version != PSM_ABI_VERSION) return fail(nullptr, PSM_ERR_ARG, "psm_config.abi_version mismatch");
  if (cfg->variant < 0 || cfg->variant > 2) return fail(nullptr, PSM_ERR_ARG, "unknown variant");
  if (cfg->block != 128) return fail(nullptr, PSM_ERR_UNSUPPORTED, "block must be 128 (the only block edge the reference uses: python_module.py:303, entry_point.py --shape 128)");
  if (cfg->c_in < 1 || cfg->c_in > 4) return fail(nullptr, PSM_ERR_ARG, "c_in must be 1..4");
  if (cfg->c_out < 1 || cfg->c_out > 2) return fail(nullptr, PSM_ERR_ARG, "c_out must be 1 or 2");
  if (cfg->variant == PSM_VARIANT_GRADP && cfg->c_out != 2) return fail(nullptr, PSM_ERR_ARG, "gradp needs c_out == 2");
  if (cfg->variant != PSM_VARIANT_GRADP && cfg->c_out != 1) return fail(nullptr, PSM_ERR_ARG, "this variant needs c_out == 1");
  if (cfg->p_in < 1 || cfg->p_in > 1024 || cfg->p_out < 1 || cfg->p_out > 1024) return fail(nullptr, PSM_ERR_ARG, "p_in/p_out must be 1..1024");
  if (cfg->n_dense < 1 || cfg->n_dense > 64) return fail(nullptr, PSM_ERR_ARG, "n_dense must be 1..64");
  if (cfg->scaler < 0 || cfg->scaler > 2) return fail(nullptr, PSM_ERR_ARG, "Standardization method not valid");
  if (cfg->sdf_channel < 0 || cfg->sdf_channel >= cfg->c_in) return fail(nullptr, PSM_ERR_ARG, "sdf_channel outside the input channels");
  if (cfg->max_cases < 1) return fail(nullptr, PSM_ERR_ARG, "max_cases must be >= 1");
  if (cfg->overlap < 0 || cfg->overlap >= cfg->block) return fail(nullptr, PSM_ERR_ARG, "overlap must lie in [0, block)");
  if (cfg->precision != PSM_PRECISION_F32 && cfg->precision != PSM_PRECISION_BF16) return fail(nullptr, PSM_ERR_ARG, "unknown precision");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, PSM_ERR_NO_DEVICE, "no HIP device: the surrogate path has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PSM_ERR_ARG, "device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, PSM_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, PSM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  psm_handle* h = new psm_handle();
  h->cfg = *cfg;
  h->S = cfg->block;
  h->ov = cfg->overlap > 0 ? cfg->overlap : psm_default_overlap(cfg->variant, cfg->block);
  h->K_in = h->S * h->S * cfg->c_in;
  h->K_out = h->S * h->S * cfg->c_out;
  h->ld_in = round_up(cfg->p_in, 32);
  h->ld_out = round_up(cfg->p_out, 32);
  h->NT = h->ld_in / 32;
  h->n_slices = h->S * h->S / PSM_PIX_PER_SLICE;
  h->Gd = h->ld_out / 8;
  h->n_coltiles = h->K_out / 32;
  h->dense.resize(cfg->n_dense);
  // Launch mode: plain stream launches by default.  Measured on MI355X (ROCm 7.2) one hipGraph
  // replay per solve costs ~5 us more per solve than the same kernels launched eagerly (a gap
  // of ~8 us between consecutive replays against back-to-back kernels), and the host enqueues
  // the ~9 launches faster than the GPU retires them.  PSM_GRAPH=1 selects graph replay.
  const char* ug = getenv("PSM_GRAPH");
  h->use_graph = (ug && ug[0] == '1');
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(nullptr, PSM_ERR_HIP, "cannot create a stream on the device");
  }
  for (int i = 0; i < psm_handle::RING; ++i) (void)hipEventCreateWithFlags(&h->scale_ev[i], hipEventDisableTiming);
  // guard of the bound-geometry contract: one word per workspace in mapped pinned memory (host-side detection; without a
  // mapped view the device-side NaN poisoning still works) and the zero the unguarded solves read as their flag
  {
    const char* ng = getenv("PSM_NO_GUARD");
    h->guard_on = !(ng && ng[0] == '1');
    h->x6_mode = getenv("PSM_X6") ? atoi(getenv("PSM_X6")) : -1;
    if (hipHostMalloc((void**)&h->h_guard, 64 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
      memset(h->h_guard, 0, 64 * sizeof(int));
      if (hipHostGetDevicePointer((void**)&h->m_guard, h->h_guard, 0) != hipSuccess) { (void)hipGetLastError(); h->m_guard = nullptr; }
    } else { (void)hipGetLastError(); h->h_guard = nullptr; }
    h->ws0.gidx = 0;
    for (int i = 0; i < psm_handle::SLOTS; ++i) h->slot[i].ws.gidx = 1 + i;
    if (psm_dev_malloc((void**)&h->d_gzero, sizeof(float)) != hipSuccess || hipMemset(h->d_gzero, 0, sizeof(float)) != hipSuccess) {
      psm_destroy(h);
      return fail(nullptr, PSM_ERR_NOMEM, "hipMalloc failed");
    }
  }
  *out = h;
  return PSM_OK;
}


void psm_destroy(psm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)hipDeviceSynchronize();
  free_plan(h);
  free_geometry(h);
  for (auto& d : h->dense) { dev_free(d.W); dev_free(d.b); if (d.Wp) { (void)psm_dev_free(d.Wp); d.Wp = nullptr; } }
  dev_free(h->d_mean_in); dev_free(h->d_mean_out); dev_free(h->d_bpack_in); dev_free(h->d_bpack_out); dev_free(h->d_bpack_x6);
  if (h->scr_dev) (void)psm_dev_free(h->scr_dev);
  if (h->scr_pin) (void)hipHostFree(h->scr_pin);
  dev_free(h->d_comp_nat); dev_free(h->d_g2); dev_free(h->d_c2); dev_free(h->d_cnt); dev_free(h->d_row_of); dev_free(h->d_ownbits);
  dev_free(h->d_ia); dev_free(h->d_ib); dev_free(h->d_sa); dev_free(h->d_sb);
  dev_free(h->d_maskbits); dev_free(h->d_gzero);
  dev_free(h->d_g2p); dev_free(h->d_c2p); dev_free(h->d_cntp); dev_free(h->d_cfa0); dev_free(h->d_row_of_p);
  for (auto& c : h->conv1d) { dev_free(c.W); dev_free(c.b); }
  for (auto& d : h->dense) { dev_free(d.ln_gamma); dev_free(d.ln_beta); }
  if (h->h_guard) (void)hipHostFree(h->h_guard);
  for (int i = 0; i < psm_handle::RING; ++i) {
    if (h->h_scale[i]) (void)hipHostFree(h->h_scale[i]);
    if (h->scale_ev[i]) (void)hipEventDestroy(h->scale_ev[i]);
  }
  for (auto& p : h->timed_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (auto& r : h->host_regs) (void)hipHostUnregister(r.base);
  delete h;
}


int psm_set_pca(psm_handle* h, const double* comp_in, const double* mean_in, const double* comp_out, const double* mean_out) {
  if (!h) return PSM_ERR_ARG;
  if (!comp_in || !mean_in || !comp_out || !mean_out) return fail(h, PSM_ERR_ARG, "null PCA array");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  destroy_graphs(h);
  h->bound = false;
  std::vector<float> mi(h->K_in), mo(h->K_out);
  for (int k = 0; k < h->K_in; ++k) mi[k] = (float)mean_in[k];
  for (int k = 0; k < h->K_out; ++k) mo[k] = (float)mean_out[k];
  int rc;
  if ((rc = dev_upload(h, &h->d_mean_in, mi))) return rc;
  if ((rc = dev_upload(h, &h->d_mean_out, mo))) return rc;
  if (h->cfg.precision == PSM_PRECISION_BF16) {
    std::vector<uint16_t> bi = pack_comp_in_bf16(comp_in, h->cfg.p_in, h->K_in, h->cfg.c_in, h->S, h->NT);
    std::vector<uint16_t> bo = pack_comp_out_bf16(comp_out, h->cfg.p_out, h->K_out, h->ld_out / 16);
    uint16_t *di = nullptr, *dox = nullptr;
    if ((rc = dev_upload(h, &di, bi))) return rc;
    if ((rc = dev_upload(h, &dox, bo))) { dev_free(di); return rc; }
    dev_free(h->d_bpack_in); dev_free(h->d_bpack_out);
    h->d_bpack_in = reinterpret_cast<float4*>(di);
    h->d_bpack_out = reinterpret_cast<float4*>(dox);
    {                                                    // natural-layout copy of the ROUNDED basis (psm_bind_geometry)
      std::vector<float> nat((size_t)h->ld_out * h->K_out, 0.f);
      for (int p = 0; p < h->cfg.p_out; ++p)
        for (int k = 0; k < h->K_out; ++k) {
          const uint32_t u = (uint32_t)f2bf(comp_out[(int64_t)p * h->K_out + k]) << 16;
          float v; memcpy(&v, &u, 4);
          nat[(size_t)p * h->K_out + k] = v;
        }
      if ((rc = dev_upload(h, &h->d_comp_nat, nat))) return rc;
    }
  } else {
  if ((rc = dev_upload(h, &h->d_bpack_in, pack_comp_in(comp_in, h->cfg.p_in, h->K_in, h->cfg.c_in, h->S, h->NT)))) return rc;
  dev_free(h->d_bpack_x6);                                  // the pre-split copy of the large-batch encode is rebuilt on first use
  if ((rc = dev_upload(h, &h->d_bpack_out, pack_comp_out(comp_out, h->cfg.p_out, h->K_out, h->Gd)))) return rc;
  {
    std::vector<float> nat((size_t)h->ld_out * h->K_out, 0.f);
    for (int p = 0; p < h->cfg.p_out; ++p)
      for (int k = 0; k < h->K_out; ++k) nat[(size_t)p * h->K_out + k] = (float)comp_out[(int64_t)p * h->K_out + k];
    if ((rc = dev_upload(h, &h->d_comp_nat, nat))) return rc;
  }
  }
  h->have_pca = true;
  return PSM_OK;
}


int psm_set_dense(psm_handle* h, int32_t layer, int32_t n_in, int32_t n_out, const float* kernel, const float* bias) {
  if (!h) return PSM_ERR_ARG;
  if (layer < 0 || layer >= (int)h->dense.size()) return fail(h, PSM_ERR_ARG, "layer index out of range");
  const int in_cap = h->conv1d.empty() ? 4096 : (1 << 18);
  if (!kernel || !bias || n_in < 1 || n_out < 1 || n_in > (layer == 0 ? in_cap : 4096) || n_out > 4096) return fail(h, PSM_ERR_ARG, "bad dense layer");
  const int first_in = h->conv1d.empty() ? h->cfg.p_in : h->cfg.p_in * h->conv1d.back().cout;     // Flatten of [p_in, filters]
  if (layer == 0 && n_in != first_in)
    return fail(h, PSM_ERR_ARG, h->conv1d.empty() ? "first layer input width must equal p_in" : "first Dense layer after the Conv1D stack must take p_in * filters inputs (Flatten)");
  if (layer == (int)h->dense.size() - 1 && n_out != h->cfg.p_out) return fail(h, PSM_ERR_ARG, "head width must equal p_out");
  if (layer > 0 && h->dense[layer - 1].set && h->dense[layer - 1].n_out != n_in) return fail(h, PSM_ERR_ARG, "dense layers do not chain");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  destroy_graphs(h);
  h->bound = false;
  DenseLayer& d = h->dense[layer];
  d.linear = false;                                     // psm_set_attention sets it again after this call
  if (d.ln && (d.n_out != n_out || (d.ln_residual && n_in != n_out))) {      // a LayerNormalization of another width, or its residual x + input on a
    d.ln = false; d.ln_residual = false; dev_free(d.ln_gamma); dev_free(d.ln_beta);   // layer that is no longer square: set it again after this call
  }
  d.n_in = n_in; d.n_out = n_out; d.Kpad = round_up(n_in, 32); d.ldw = round_up(n_out, 32);
  std::vector<float> W((size_t)d.Kpad * d.ldw, 0.f), b(d.ldw, 0.f);
  for (int k = 0; k < n_in; ++k) memcpy(&W[(size_t)k * d.ldw], kernel + (size_t)k * n_out, n_out * sizeof(float));
  memcpy(b.data(), bias, n_out * sizeof(float));
  int rc;
  // MFMA-packed copy: Wp[nt][kg][lane][j] = W[16*kg + 4*(lane>>4) + j][16*nt + (lane&15)], contraction
  // padded with zero rows to Kp (128, 256 or a multiple of 512: whole passes of the dense kernel)
  d.Kp = d.Kpad <= 128 ? 128 : (d.Kpad <= 256 ? 256 : round_up(d.Kpad, 512));
  const int groups = d.Kp / 16, ntiles = d.ldw / 16;
  std::vector<float> Wp((size_t)ntiles * groups * 64 * 4, 0.f);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int kg = 0; kg < groups; ++kg)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int k = 16 * kg + 4 * (lane >> 4) + j, n = 16 * nt + (lane & 15);
          if (k < n_in && n < n_out) Wp[(((size_t)nt * groups + kg) * 64 + lane) * 4 + j] = kernel[(size_t)k * n_out + n];
        }
  if (d.Wp) { (void)psm_dev_free(d.Wp); d.Wp = nullptr; }
  if (h->cfg.precision == PSM_PRECISION_BF16) {
    std::vector<uint16_t> Wb(W.size()), Wpb(Wp.size());
    for (size_t q = 0; q < W.size(); ++q) Wb[q] = f2bf(W[q]);
    for (size_t q = 0; q < Wp.size(); ++q) Wpb[q] = f2bf(Wp[q]);
    uint16_t *dw = nullptr, *dwp = nullptr;
    if ((rc = dev_upload(h, &dw, Wb))) return rc;
    if ((rc = dev_upload(h, &dwp, Wpb))) { dev_free(dw); return rc; }
    dev_free(d.W);
    d.W = reinterpret_cast<float*>(dw);
    d.Wp = dwp;
  } else {
    if ((rc = dev_upload(h, &d.W, W))) return rc;
    float* dwp = nullptr;
    if ((rc = dev_upload(h, &dwp, Wp))) return rc;
    d.Wp = dwp;
  }
  if ((rc = dev_upload(h, &d.b, b))) return rc;
  d.set = true;
  return PSM_OK;
}


int psm_set_attention(psm_handle* h, int32_t layer, int32_t d_model, int32_t n_heads, int32_t value_dim, const float* Wv, const float* bv,
                      const float* Wo, const float* bo) {
  if (!h) return PSM_ERR_ARG;
  if (layer < 1 || layer >= (int)h->dense.size() - 1) return fail(h, PSM_ERR_ARG, "the attention block must sit between the first Dense layer and the head");
  if (!Wv || !bv || !Wo || !bo || d_model < 1 || d_model > 4096 || n_heads < 1 || value_dim < 1 || (int64_t)n_heads * value_dim > 65536)
    return fail(h, PSM_ERR_ARG, "bad attention block");
  // Sequence length 1 (NNs.py:54 tf.expand_dims(x, 1), NNs.py:55 attention of x with itself): the softmax over the single key
  // is exactly 1 whatever the query and key projections give, so the block is value projection -> output projection:
  //   out = (x . Wv + bv) . Wo + bo = x . (Wv Wo) + (bv Wo + bo)      -- folded here in float64, one Dense launch without ReLU
  const int HV = n_heads * value_dim;
  std::vector<double> W((size_t)d_model * d_model, 0.0), b(d_model, 0.0);
  for (int i = 0; i < d_model; ++i)
    for (int k = 0; k < HV; ++k) {
      const double v = Wv[(size_t)i * HV + k];
      const float* wo = Wo + (size_t)k * d_model;
      double* wr = &W[(size_t)i * d_model];
      for (int j = 0; j < d_model; ++j) wr[j] += v * (double)wo[j];
    }
  for (int j = 0; j < d_model; ++j) b[j] = bo[j];
  for (int k = 0; k < HV; ++k)
    for (int j = 0; j < d_model; ++j) b[j] += (double)bv[k] * (double)Wo[(size_t)k * d_model + j];
  std::vector<float> Wf(W.begin(), W.end()), bf(b.begin(), b.end());
  int rc = psm_set_dense(h, layer, d_model, d_model, Wf.data(), bf.data());
  if (rc) return rc;
  h->dense[layer].linear = true;
  return PSM_OK;
}


int psm_set_layernorm(psm_handle* h, int32_t layer, int32_t n, const float* gamma, const float* beta, float epsilon, int32_t residual) {
  if (!h) return PSM_ERR_ARG;
  if (layer < 0 || layer >= (int)h->dense.size() - 1) return fail(h, PSM_ERR_ARG, "LayerNormalization follows a hidden layer (not the head)");
  DenseLayer& d = h->dense[layer];
  if (!d.set) return fail(h, PSM_ERR_STATE, "set the Dense layer (psm_set_dense / psm_set_attention) before its LayerNormalization");
  if (!gamma || !beta || n != d.n_out) return fail(h, PSM_ERR_ARG, "LayerNormalization width must equal the layer's output width");
  if (!(epsilon > 0.f)) return fail(h, PSM_ERR_ARG, "epsilon must be positive");
  if (residual && d.n_in != d.n_out) return fail(h, PSM_ERR_ARG, "the residual x + input needs a square layer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  destroy_graphs(h);
  h->bound = false;
  // zero-padded to whole 16-byte pieces past the consumer's leading dimension: a Dense launch that applies this normalisation to
  // its input (launch_all) reads gamma / beta with the clamped column index of its operand loads
  std::vector<float> g(round_up(n, 32) + 32, 0.f), b(round_up(n, 32) + 32, 0.f);
  std::copy(gamma, gamma + n, g.begin()); std::copy(beta, beta + n, b.begin());
  int rc;
  if ((rc = dev_upload(h, &d.ln_gamma, g)) || (rc = dev_upload(h, &d.ln_beta, b))) return rc;
  d.ln = true; d.ln_residual = residual != 0; d.ln_eps = epsilon;
  return PSM_OK;
}


int psm_set_conv1d(psm_handle* h, int32_t layer, int32_t n_layers, int32_t kernel_size, int32_t c_in, int32_t c_out, const float* kernel,
                   const float* bias) {
  if (!h) return PSM_ERR_ARG;
  if (n_layers < 1 || n_layers > 32 || layer < 0 || layer >= n_layers) return fail(h, PSM_ERR_ARG, "Conv1D layer index out of range");
  if (!kernel || !bias || kernel_size < 1 || kernel_size > 15 || c_in < 1 || c_out < 1 || c_in > 2048 || c_out > 2048)
    return fail(h, PSM_ERR_ARG, "bad Conv1D layer");
  if (h->cfg.precision != PSM_PRECISION_F32) return fail(h, PSM_ERR_UNSUPPORTED, "the conv1D_PCA head is float32 only");
  if ((int64_t)h->cfg.p_in * c_out > (1 << 18)) return fail(h, PSM_ERR_UNSUPPORTED, "Conv1D activation wider than 2^18 per block");
  // every argument check comes BEFORE the handle is touched: a rejected call leaves graphs, binding, stack and plan as they were
  if (layer == 0 && c_in != 1) return fail(h, PSM_ERR_ARG, "the first Conv1D layer sees the coefficients as [p_in, 1]: c_in must be 1");
  const bool same_stack = (int)h->conv1d.size() == n_layers;
  if (same_stack && layer > 0 && h->conv1d[layer - 1].set && h->conv1d[layer - 1].cout != c_in) return fail(h, PSM_ERR_ARG, "Conv1D layers do not chain");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  destroy_graphs(h);
  h->bound = false;
  if (!same_stack) {
    for (auto& c : h->conv1d) { dev_free(c.W); dev_free(c.b); }
    h->conv1d.assign(n_layers, Conv1dLayer{});
  }
  // the workspaces (c1_stride, d_c1[]) are sized from p_in * c_out of every layer: a new stack, or the same stack with another
  // filter count, needs a new plan: the plan is dropped and the next solve fails with PSM_ERR_STATE until psm_plan_grid is called again
  if (h->planned && (!same_stack || !h->conv1d[layer].set || h->conv1d[layer].cout != c_out)) free_plan(h);
  Conv1dLayer& c = h->conv1d[layer];
  c.k = kernel_size; c.cin = c_in; c.cout = c_out;
  std::vector<float> W(kernel, kernel + (size_t)kernel_size * c_in * c_out), b(bias, bias + c_out);
  int rc;
  if ((rc = dev_upload(h, &c.W, W)) || (rc = dev_upload(h, &c.b, b))) return rc;
  c.set = true;
  return PSM_OK;
}


int psm_set_scaler(psm_handle* h, const double* in_a, const double* in_b, const double* out_a, const double* out_b) {
  if (!h) return PSM_ERR_ARG;
  if (!in_a || !out_a) return fail(h, PSM_ERR_ARG, "null scaler array");
  if (h->cfg.scaler != PSM_SCALER_MAX_ABS && (!in_b || !out_b)) return fail(h, PSM_ERR_ARG, "null scaler array");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  destroy_graphs(h);                          // captured launches hold the addresses of the arrays re-uploaded below
  h->bound = false;
  std::vector<float> ia(h->ld_in, 0.f), ib(h->ld_in, 0.f), sa(h->ld_out, 0.f), sb(h->ld_out, 0.f);
  // x_in = coeff*ia + ib ; res' = res*sa + sb  (affine forms of SMD:505-539, evaluated in f64 here)
  for (int p = 0; p < h->cfg.p_in; ++p) {
    double a, b;
    if (h->cfg.scaler == PSM_SCALER_MAX_ABS) { a = 1.0 / in_a[0]; b = 0.0; }
    else if (h->cfg.scaler == PSM_SCALER_STD) { a = 1.0 / in_b[p]; b = -in_a[p] / in_b[p]; }
    else { a = 1.0 / (in_b[p] - in_a[p]); b = -in_a[p] / (in_b[p] - in_a[p]); }
    ia[p] = (float)a; ib[p] = (float)b;
  }
  for (int p = 0; p < h->cfg.p_out; ++p) {
    double a, b;
    if (h->cfg.scaler == PSM_SCALER_MAX_ABS) { a = out_a[0]; b = 0.0; }
    else if (h->cfg.scaler == PSM_SCALER_STD) { a = out_b[p]; b = out_a[p]; }
    else { a = out_b[p] - out_a[p]; b = out_a[p]; }
    sa[p] = (float)a; sb[p] = (float)b;
  }
  int rc;
  if ((rc = dev_upload(h, &h->d_ia, ia))) return rc;
  if ((rc = dev_upload(h, &h->d_ib, ib))) return rc;
  if ((rc = dev_upload(h, &h->d_sa, sa))) return rc;
  if ((rc = dev_upload(h, &h->d_sb, sb))) return rc;
  h->have_scaler = true;
  return PSM_OK;
}

}  // extern "C"
