// psm_api.cpp -- C-ABI of libpsm_hip.so (include/psm.h): handle, weight packing,
// HBM layout, hipGraph-captured launch sequence, pinned host staging.
// Compiled with hipcc for gfx950 only.  There is no CPU fallback: without a
// usable device psm_create fails with PSM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include "psm_alloc.h"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/psm.h"
#include "psm_kernels.h"
#include "psm_mesh.h"
#include "psm_plan.h"

namespace {
thread_local std::string g_create_error;

// The ring keeps PSM_RING_SLOTS tickets in flight on their own streams, each stream with copy and kernel work; with the HIP
// runtime's default number of hardware queues streams share queues and neighbouring tickets end up behind each other
// (measured with 8 slots: 50 us per solve with 8 queues, 40 with 4, 34-35 with 12 / 16 / 32).  The runtime reads
// GPU_MAX_HW_QUEUES once, when it initialises: that is the HOST PROGRAM's choice (bench.py and INTEGRATION.md set 16) --
// the library never touches the environment of the process it is loaded into.

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct DenseLayer {
  int n_in = 0, n_out = 0, Kpad = 0, ldw = 0, Kp = 0;
  float* W = nullptr;      // [Kpad][ldw] (bf16 precision: bf16 elements)
  void* Wp = nullptr;      // MFMA-packed copy, see psm_dense_kernel
  float* b = nullptr;
  bool set = false;
  bool linear = false;     // hidden layer without ReLU (the folded attention block of densePCA_attention)
  // LayerNormalization behind this layer (densePCA_attention, NNs.py:56, 64): act = LN(act [+ this layer's input]) * gamma + beta
  bool ln = false, ln_residual = false;
  float *ln_gamma = nullptr, *ln_beta = nullptr;
  float ln_eps = 1e-3f;
};

struct Conv1dLayer {            // conv1D_PCA head (NNs.py:75-124)
  int k = 0, cin = 0, cout = 0;
  float *W = nullptr, *b = nullptr;
  bool set = false;
};

struct GraphKey {
  int n; const void* g; void* f;
  bool operator<(const GraphKey& o) const { return std::tie(n, g, f) < std::tie(o.n, o.g, o.f); }
};

// Everything ONE in-flight solve writes.  The handle owns one for the synchronous / device entries (ws0) and one per
// ring slot, so that the solves of neighbouring tickets run on their own streams without sharing scratch.
struct Workspace {
  float *d_part = nullptr, *d_xin = nullptr, *d_act[2] = {nullptr, nullptr}, *d_res = nullptr, *d_pred = nullptr;
  float *d_row_scale = nullptr;
  float4* d_spart = nullptr;
  float2* d_colpart = nullptr;
  float *d_offs = nullptr, *d_shift = nullptr;
  float* d_dots = nullptr;            // strip dots of the geometry-bound path (allocated by the bind)
  float* d_dots2 = nullptr;           // pair dots of the closed-form chain (allocated by the bind)
  float* d_c1[2] = {nullptr, nullptr}; // Conv1D activations of the conv1D_PCA head (ping-pong), [Mpad][c1_stride]
  float* d_gflags = nullptr;          // guard flags of the bound-geometry contract (psm_kernels.h PsmGuardArgs; allocated by the bind)
  int gidx = 0;                       // this workspace's word in the handle's mapped guard page (0 = ws0, 1 + i = ring slot i)
};
}  // namespace

struct psm_handle {
  psm_config cfg{};
  std::string err;
  int S = 0, ov = 0, K_in = 0, K_out = 0, ld_in = 0, ld_out = 0, NT = 0, n_slices = 0, Gd = 0, n_coltiles = 0;
  bool have_pca = false, have_scaler = false;
  std::vector<DenseLayer> dense;
  std::vector<Conv1dLayer> conv1d;      // in front of the dense layers when the model is the reference's conv1D_PCA
  int64_t c1_stride = 0;                // floats per block row of the Conv1D activation buffers
  float *d_mean_in = nullptr, *d_mean_out = nullptr;
  float4 *d_bpack_in = nullptr, *d_bpack_out = nullptr;
  uint4* d_bpack_x6 = nullptr;          // encode basis as three bf16 planes in MFMA fragment order (pack_comp_in_x6): the large-batch encode
  float *d_ia = nullptr, *d_ib = nullptr, *d_sa = nullptr, *d_sb = nullptr;
  // plan
  bool planned = false;
  PsmPlan plan;
  int Ny = 0, Nx = 0, B = 0, Mcap = 0, Mpad_cap = 0, n_strips = 0, Lmax = 0, max_width = 0;
  Workspace ws0;
  int64_t* d_row_base = nullptr;
  float* d_ones = nullptr;
  int32_t *d_strips = nullptr, *d_blk = nullptr, *d_owner = nullptr, *d_shiftA = nullptr, *d_shiftB = nullptr, *d_shiftOwnA = nullptr, *d_shiftOwnB = nullptr;
  float* d_shiftW = nullptr;
  PsmBlock* d_blocks = nullptr;
  int n_bands = 0;
  unsigned long long* d_stamps = nullptr;
  // mesh-side tables (psm_set_geometry)
  bool have_geometry = false, have_g2m = false;
  int64_t n_cells = 0;
  int32_t *d_vtx_m2g = nullptr, *d_src_of_cell = nullptr, *d_vtx_g2m = nullptr, *d_cell_of_point = nullptr;
  double *d_wts_m2g = nullptr, *d_sdf = nullptr, *d_wts_g2m = nullptr, *d_cells = nullptr, *d_p = nullptr, *d_umax = nullptr, *d_umax_part = nullptr;
  uint8_t* d_near_wall = nullptr;
  // U_to_gradP integration (psm_set_integration)
  bool have_integ = false;
  PsmIntegArgs integ{};
  int2 *d_fixups = nullptr, *d_pairs = nullptr;
  float *d_integ_buf = nullptr, *d_gradp = nullptr;
  double *h_cells = nullptr, *h_p = nullptr;
  const double* pinned_cells = nullptr;   // caller buffers registered with psm_pin_buffers (DMA without staging copies)
  double* pinned_p = nullptr;
  double* pinned_p_dev = nullptr;       // device-side address of the registered output (the last kernel writes p straight into it)
  const double* pinned_cells_dev = nullptr;   // device-side address of the registered input (psm_stage_cells_kernel reads it over PCIe)
  hipGraphExec_t mesh_graph = nullptr;  // psm_solve on registered buffers: stage + to_grid + the solve + to_mesh as ONE graph replay
  double maxs[4] = {1, 1, 1, 1};
  int normalise_sdf = 0, fill_input = 0;
  double case_maxs[4] = {1, 1, 1, 1}, case_delta = 5e-3, case_wall = 0.05;   // psm_set_case (PM:106-109, 195, 494)
  int case_every = 10;                                                        // PM:94-95
  float *d_grid_stage = nullptr, *d_fields_stage = nullptr;
  float *h_grid = nullptr, *h_fields = nullptr;
  // host-buffer submission ring (psm_submit_grid / psm_wait_grid): pinned in/out + device in/out per slot
  // One ring slot = pinned host buffers + device buffers + its own workspace, stream and graphs: the H2D copy, the
  // kernels and the D2H copy of a ticket run in order on the slot's stream, different slots overlap freely.
  struct Slot {
    float *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr, *h_rs = nullptr;
    float *m_in = nullptr, *m_out = nullptr, *m_rs = nullptr;   // device-side addresses of the pinned buffers (mapped)
    Workspace ws;
    const float* last_src = nullptr;   // what the ticket in flight was launched with (re-run on the general path when
    float* last_dst = nullptr;         // the guard of the bound-geometry contract trips)
    std::vector<float> last_scale;
    hipStream_t st = nullptr;
    hipEvent_t ev_out = nullptr;
    hipGraphExec_t g_full = nullptr, g_kern = nullptr;   // H2D + kernels + D2H on the slot's own buffers / the kernels alone
    int g_full_key = -1, g_kern_key = -1;
    int state = 0;             // 0 free, 1 acquired (the caller is packing), 2 in flight
    int64_t ticket = -1;
    int n_cases = 0;
    float* user_out = nullptr; // where psm_wait_grid copies to when the caller gave the pointer at submission
    bool direct_out = false;   // the D2H went straight into user_out (registered memory)
  };
  static constexpr int SLOTS = PSM_RING_SLOTS;
  Slot slot[SLOTS];
  bool ring_ready = false;
  int ring_slots = SLOTS;      // slots in rotation (PSM_RING_USE=n, n <= PSM_RING_SLOTS: experiments)
  int ring_graph = 1;          // PSM_RING_GRAPH=0: plain launches on the slot streams
  int ring_dma = 1;            // PSM_RING_PULL=1 clears it: the GPU pulls the grid from / stores the field to the mapped pinned
                               // buffers itself instead of hipMemcpyAsync (SDMA) copies around the kernels -- measured slower
  int64_t next_ticket = 0;
  struct HostReg { char* base; size_t bytes; char* dev; };
  std::vector<HostReg> host_regs;                    // psm_host_register
  // row-scale upload ring (pinned)
  static constexpr int RING = 8;
  float* h_scale[RING] = {};
  hipEvent_t scale_ev[RING] = {};
  int scale_pos = 0;
  hipStream_t stream = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;
  bool use_graph = true;
  bool fused_assemble = false;
  // scratch of the helper entries (gaussian filter, mesh -> grid, Poisson features, gradp integration): one device and one
  // pinned host buffer, grown on demand and reused -- a hipMalloc / hipFree pair per call cost more than the kernels
  void *scr_dev = nullptr, *scr_pin = nullptr;
  size_t scr_dev_cap = 0, scr_pin_cap = 0;
  // geometry-bound fast path (psm_bind_geometry): tables of psm_kernels.h PsmBindArgs
  bool bound = false, bound_zero_fill = false;
  int bound_scope = 0;                  // 2: every single-case solve (psm_bind_geometry); 1: psm_solve only (bound by psm_set_geometry)
  bool in_mesh_solve = false;
  bool mesh_inflight = false;           // psm_solve_begin enqueued, psm_solve_end not yet called
  double* mesh_copy_out = nullptr;      // where psm_solve_end copies p to (null: it was DMA'd / stored into the caller's registered array)
  int bound_rows = 0;                   // table rows per case
  int bound_cases = 0;                  // cases bound (solves with exactly this many cases take the bound path)
  float *d_comp_nat = nullptr;          // comp_out in natural layout [ld_out][K_out] (f32 precision only)
  float *d_g2 = nullptr, *d_c2 = nullptr, *d_cnt = nullptr;
  size_t bound_dots = 0;                // floats of Workspace::d_dots
  std::vector<uint8_t> bound_mask;      // [bound_cases][Ny*Nx] flow-cell pattern that was bound (psm_bound_mask)
  int32_t* d_row_of = nullptr;
  uint32_t* d_ownbits = nullptr;
  // closed form of the offset chain for case batches (psm_kernels.h PsmBoundBatchArgs): pair tables
  int x6_mode = -1;                     // PSM_X6 at psm_create: -1 default (see launch_all), bit 0 encode, bit 1 bound decode
  bool bound_cf = false;
  size_t cf_rows_all = 0;               // cases * c_out * B * B
  float *d_g2p = nullptr, *d_c2p = nullptr, *d_cntp = nullptr, *d_cfa0 = nullptr;
  int32_t* d_row_of_p = nullptr;
  std::vector<float> h_shiftW;          // host copy of d_shiftW [c_out][B]
  const float* last_row_scale = nullptr;   // row scale of the last solve on ws0 (introspection)
  bool last_used_cf = false;            // the last solve on ws0 took the closed form: offsets / shift are computed on demand
  // guard of the bound-geometry contract (psm_kernels.h PsmGuardArgs)
  unsigned long long* d_maskbits = nullptr;   // bound flow-cell pattern, one 64-pixel ballot per word
  int guard_ballots = 0, guard_waves = 0;
  bool guard_on = true;                 // PSM_NO_GUARD=1 switches the riders off (diagnostic)
  int *h_guard = nullptr, *m_guard = nullptr;   // mapped pinned page: one word per workspace, raised by a guard wave on mismatch
  float* d_gzero = nullptr;             // one zero: the flags of solves without a guard
  int64_t guard_trips = 0;
  int debug_skip = 0;                   // PSM_DEBUG_SKIP bit mask of kernel groups NOT launched (timing experiments only)
  bool fuse_reduce_dense1 = true;       // PSM_NO_FUSED_REDUCE=1 disables
  int last_cases = 0;
  bool last_on_ws0 = false;             // the most recent solve ran on the handle's own workspace (not a ring slot's): what psm_block_error decodes
  // event timing of one kernel group
  int timed_kernel = -1;
  int timed_repeat = 1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed_events;
  double timed_total_ms = 0.0;
  int64_t timed_launches = 0;
};

namespace {

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
double spin_budget_us() { static const double v = [] { const char* e = getenv("PSM_SPIN_US"); return e ? atof(e) : 300.0; }(); return v; }
template <typename Query, typename Block>
hipError_t bounded_wait(Query query, Block block) {
  if (sync_blocks()) return block();
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e;
  int n = 0;
  while ((e = query()) == hipErrorNotReady) {
    if ((++n & 63) == 0 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_budget_us())
      return block();
  }
  return e;
}
hipError_t wait_stream(hipStream_t st) {
  return bounded_wait([&] { return hipStreamQuery(st); }, [&] { return hipStreamSynchronize(st); });
}
hipError_t wait_event(hipEvent_t ev) {
  return bounded_wait([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); });
}

int fail(psm_handle* h, int code, const std::string& msg);
// carve helpers: 256-byte aligned pieces of the two scratch buffers
struct Carver {
  char* base; size_t off = 0;
  template <typename T> T* take(size_t n) { T* p = reinterpret_cast<T*>(base + off); off += (n * sizeof(T) + 255) & ~(size_t)255; return p; }
};
inline size_t carve_size(std::initializer_list<size_t> bytes) { size_t t = 0; for (size_t b : bytes) t += (b + 255) & ~(size_t)255; return t; }
int scratch_reserve(psm_handle* h, size_t dev_bytes, size_t pin_bytes);

int fail(psm_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail((h), PSM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

template <typename T>
int dev_alloc(psm_handle* h, T** p, size_t n) {
  if (*p) { (void)psm_dev_free(*p); *p = nullptr; }
  if (n == 0) n = 1;
  hipError_t e = psm_dev_malloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) return fail(h, PSM_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return PSM_OK;
}

template <typename T>
int dev_upload(psm_handle* h, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(h, p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(h, psm_copy_h2d(*p, v.data(), v.size() * sizeof(T)));
  return PSM_OK;
}

template <typename T>
void dev_free(T*& p) { if (p) { (void)psm_dev_free(p); p = nullptr; } }

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

void ring_drop_graphs(psm_handle* h);
void destroy_graphs(psm_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  if (h->mesh_graph) { (void)hipGraphExecDestroy(h->mesh_graph); h->mesh_graph = nullptr; }
  ring_drop_graphs(h);
}

void ws_free(Workspace& w) {
  dev_free(w.d_part); dev_free(w.d_xin); dev_free(w.d_act[0]); dev_free(w.d_act[1]); dev_free(w.d_res); dev_free(w.d_pred);
  dev_free(w.d_row_scale); dev_free(w.d_spart); dev_free(w.d_colpart); dev_free(w.d_offs); dev_free(w.d_shift); dev_free(w.d_dots);
  dev_free(w.d_gflags); dev_free(w.d_c1[0]); dev_free(w.d_c1[1]); dev_free(w.d_dots2);
}

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
  if (h->bound && h->bound_dots) { if ((rc = dev_alloc(h, &w.d_dots, h->bound_dots))) return rc; }
  if (h->bound && (rc = ws_alloc_guard(h, w))) return rc;
  if (h->bound && h->bound_cf) { if ((rc = dev_alloc(h, &w.d_dots2, h->cf_rows_all))) return rc; }
  for (int q = 0; q < 2 && h->c1_stride; ++q) {             // padding columns of the last layer's rows are read by the dense kernel
    if ((rc = dev_alloc(h, &w.d_c1[q], (size_t)h->Mpad_cap * h->c1_stride))) return rc;
    HIPCHK(h, hipMemset(w.d_c1[q], 0, (size_t)h->Mpad_cap * h->c1_stride * sizeof(float)));
  }
  return PSM_OK;
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

inline uint16_t f2bf(double v) {             // round-to-nearest-even float -> bf16 (NaN stays NaN)
  const float f = (float)v;
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
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

// ---- the launch sequence -------------------------------------------------------
// launches of a kernel group: once, or `timed_repeat` times back to back between the two timing
// events when that group is being timed (the group is idempotent; amortises the ~2.7 us an event
// pair adds to a single launch)
#define PSM_REPEAT(h, k) for (int rep_ = 0, nrep_ = (((h)->debug_skip >> (k)) & 1) ? 0 : ((h)->timed_kernel == (k) ? (h)->timed_repeat : 1); rep_ < nrep_; ++rep_)

struct Timer {                      // optional event pair around one kernel group
  psm_handle* h; hipStream_t st; int k; hipEvent_t* ev;   // ev: [PSM_K_COUNT+1] profile events or null
  void before(int kernel) {
    if (ev && kernel == 0) (void)hipEventRecord(ev[0], st);
    if (h->timed_kernel == kernel && !(kernel == PSM_K_ENCODE && !ev)) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a, st);
      h->timed_events.push_back({a, b});
    }
  }
  void after(int kernel) {
    if (ev) (void)hipEventRecord(ev[kernel + 1], st);
    if (h->timed_kernel == kernel && !(kernel == PSM_K_ENCODE && !ev)) (void)hipEventRecord(h->timed_events.back().second, st);
  }
};

// K groups of the large-batch encode (psm_encode_x6_mt_kernel) for Mpad block rows, 1 = the one-slab-per-slice forms.
// From 432 block rows up (>= 48 cases of 9 blocks): about 512 workgroups = two per CU, i.e. 512 / row groups K groups (64 cases:
// nine row groups of 64 -> 56 groups of 4-5 slices, 16.5 MB of slabs; the one-slab-per-slice form writes 75 MB).  PSM_ENCODE_KGROUPS=n forces the group count (1: the old form).
int encode_groups(const psm_handle* h, int Mpad) {
  // crossover measured on one box (us per step, whole solve): 40 cases 101.6 one-slab-per-slice / 106.2 M-tiled, 48 cases 113.8 / 110.6,
  // 56 cases 124.7 / 104.8, 64 cases 141.5 / 123.5 -- from 432 block rows (48 cases of 9 blocks) up
  static const int min_rows = getenv("PSM_ENCODE_MT_MIN_ROWS") ? atoi(getenv("PSM_ENCODE_MT_MIN_ROWS")) : 432;
  if (h->cfg.precision == PSM_PRECISION_BF16 || h->NT > 4 || Mpad % 32 != 0 || Mpad < min_rows || ((PSM_PIX_PER_SLICE * h->cfg.c_in) % 32) != 0) return 1;
  if (h->x6_mode >= 0 && !(h->x6_mode & 1)) return 1;
  static const int kg_env = getenv("PSM_ENCODE_KGROUPS") ? atoi(getenv("PSM_ENCODE_KGROUPS")) : 0;
  const int row_groups = (Mpad + PSM_ENC_MT_ROWS - 1) / PSM_ENC_MT_ROWS;
  // measured at 64 cases (one box, us: encode + reduce): 512 workgroups 49.1 + 6.2, 768: 61.4 + 8.2, 1024: 56.6 + 10.1, 1536: 56.2 + 12.9
  static const int wg_target = getenv("PSM_ENCODE_WGS") ? atoi(getenv("PSM_ENCODE_WGS")) : 512;
  int groups = std::max((h->n_slices + 7) / 8, std::min(h->n_slices, wg_target / row_groups));
  if (kg_env > 0) groups = kg_env;
  return (groups > 1 && groups <= h->n_slices && (h->n_slices + groups - 1) / groups <= 8) ? groups : 1;
}
// What that encode needs beyond the plan, built on first use and OUTSIDE any stream capture (it allocates): the basis pre-split
// into three bf16 planes (1.5 x the bytes of the float32 pack), made on the device from the float32 pack.
int ensure_encode_aux(psm_handle* h, int n_cases) {
  const int Mpad = round_up(n_cases * h->B, 32);
  if (h->d_bpack_x6 || encode_groups(h, Mpad) <= 1) return PSM_OK;
  const size_t n16 = (size_t)h->n_slices * h->NT * (PSM_PIX_PER_SLICE * h->cfg.c_in / 16) * 3 * 64;
  int rc = dev_alloc(h, &h->d_bpack_x6, n16);
  if (rc) return rc;
  HIPCHK(h, psm_launch_split_basis(h->d_bpack_in, h->d_bpack_x6, h->n_slices, h->NT, PSM_PIX_PER_SLICE * h->cfg.c_in, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PSM_OK;
}

int launch_all(psm_handle* h, Workspace& w, const float* d_grid, int n_cases, float* d_fields, const float* d_row_scale,
               hipStream_t st, hipEvent_t* prof) {
  const int M = n_cases * h->B, Mpad = round_up(M, 32);
  Timer tm{h, st, 0, prof};
  h->last_on_ws0 = (&w == &h->ws0);
  const bool bf16 = (h->cfg.precision == PSM_PRECISION_BF16);
  // geometry-bound fast path: one case, nothing but the encode group being timed / skipped
  const bool use_bound = h->bound && (h->bound_scope == 2 || h->in_mesh_solve) && n_cases == h->bound_cases && (h->timed_kernel < 0 || h->timed_kernel == PSM_K_ENCODE) && h->debug_skip == 0;
  PsmEncodeArgs ea{};
  ea.grid = d_grid; ea.mean = h->d_mean_in; ea.bpack = h->d_bpack_in; ea.part = w.d_part;
  ea.row_base = h->d_row_base; ea.row_stride = (int64_t)h->Nx * h->cfg.c_in;
  ea.M = M; ea.Mpad = Mpad; ea.NT = h->NT; ea.ldp = h->ld_in; ea.S = h->S; ea.c_in = h->cfg.c_in;
  bool aligned = ((h->Nx * h->cfg.c_in) % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_grid) & 15) == 0) &&
                 ((h->Ny * (int64_t)h->Nx * h->cfg.c_in) % 4 == 0);
  for (auto& b : h->plan.blocks) if ((b.x0 * h->cfg.c_in) % 4 != 0) aligned = false;
  ea.aligned = aligned ? 1 : 0;
  { static const bool chunked = getenv("PSM_ENCODE_CHUNKED") != nullptr; ea.whole = chunked ? 0 : 1; }
  // arithmetic of the encode contraction: exact-float32 MFMA for a single row tile (one case: the launch is bound by the basis
  // stream, the matrix phase is short), the x6 form (six bf16 MFMA terms of exactly split operands, float32 accuracy,
  // psm_encode_x6_kernel) from two row tiles up, where the matrix phase is the longest serial phase of the launch
  // (8 cases: 17.6 -> 15.5 us, 64 cases: 75 -> 60 us).  PSM_X6=0 / 1 forces float32 / x6 everywhere.
  ea.x6 = h->x6_mode < 0 ? (Mpad > 32 ? 1 : 0) : ((h->x6_mode & 1) ? 1 : 0);
  // Large case batches (>= 32 cases of 9 blocks): the M-tiled, wave-specialised x6 form (encode_groups / ensure_encode_aux)
  ea.kgroup = 1;
  int n_slabs = h->n_slices;
  {
    const int groups = (ea.x6 && !bf16) ? encode_groups(h, Mpad) : 1;
    if (groups > 1 && h->d_bpack_x6) { ea.kgroup = groups; n_slabs = groups; ea.bpack_x6 = h->d_bpack_x6; }
  }

  if (h->timed_kernel == PSM_K_ENCODE && !prof) {
    // dominant kernel: dispatch-level begin / end stamps (no marker packets around the launch)
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    h->timed_events.push_back({e0, e1});
    HIPCHK(h, bf16 ? psm_launch_encode_bf16(ea, st, e0, e1) : psm_launch_encode(ea, st, e0, e1));
  } else {
    tm.before(PSM_K_ENCODE);
    PSM_REPEAT(h, PSM_K_ENCODE) HIPCHK(h, bf16 ? psm_launch_encode_bf16(ea, st) : psm_launch_encode(ea, st));
    tm.after(PSM_K_ENCODE);
  }

  // bound-geometry contract: guard riders in the launch that computes the strip dots (not for psm_solve, whose grid is
  // built from the bound sdfunct itself)
  PsmGuardArgs ga{};
  const bool guard = use_bound && h->guard_on && h->bound_scope == 2 && h->d_maskbits && w.d_gflags;
  if (guard) {
    ga.sdf = d_grid + h->cfg.sdf_channel; ga.bits = h->d_maskbits; ga.flags = w.d_gflags;
    ga.host_flag = h->m_guard ? h->m_guard + w.gidx : nullptr;
    ga.npix = (long long)n_cases * h->Ny * h->Nx; ga.c_in = h->cfg.c_in; ga.n_ballots = h->guard_ballots; ga.n_waves = h->guard_waves;
  }
  const float* gflags = guard ? w.d_gflags : h->d_gzero;
  const int n_gwaves = guard ? h->guard_waves : 1;
  // case batches (and single cases of more than 64 blocks) on a bound geometry: closed form of the chain where it was built
  const bool use_cf = use_bound && h->bound_cf && w.d_dots2;
  const int CB = h->cfg.c_out * h->B;
  if (&w == &h->ws0) { h->last_row_scale = d_row_scale; h->last_used_cf = use_cf; }
  PsmReduceArgs ra{w.d_part, w.d_xin, h->d_ia, h->d_ib, n_slabs, Mpad, h->ld_in};
  const int nl = (int)h->dense.size();
  auto dense_args = [&](int l, const float* cur, int ld_cur) {
    const DenseLayer& d = h->dense[l];
    const bool head = (l == nl - 1);
    PsmDenseArgs da{};
    da.in = cur; da.ld_in = ld_cur; da.W = d.W; da.ld_w = d.ldw; da.bias = d.b; da.Wp = d.Wp; da.Kp = d.Kp;
    da.sa = h->d_sa; da.sb = h->d_sb;
    da.out = head ? w.d_res : w.d_act[l & 1]; da.ld_out = d.ldw;
    da.Kpad = d.Kpad; da.Mpad = Mpad; da.relu = (head || d.linear) ? 0 : 1; da.head = head ? 1 : 0;
    da.bf16 = (h->cfg.precision == PSM_PRECISION_BF16) ? 1 : 0;
    da.layer = l;
    return da;
  };
  // few block rows: slab reduce + first dense layer in one launch (one workgroup per row)
  const bool c1 = !h->conv1d.empty();
  const bool fuse1 = h->fuse_reduce_dense1 && Mpad <= 128 && h->ld_in <= 512 && h->dense[0].ldw <= 1024 &&
                     h->timed_kernel != PSM_K_REDUCE && !c1;
  int l_first = 0;
  if (fuse1) {
    tm.before(PSM_K_REDUCE);
    tm.after(PSM_K_REDUCE);
    tm.before(PSM_K_MLP);
  } else {
    tm.before(PSM_K_REDUCE);
    PSM_REPEAT(h, PSM_K_REDUCE) HIPCHK(h, psm_launch_reduce(ra, st));
    tm.after(PSM_K_REDUCE);
    tm.before(PSM_K_MLP);
  }
  PSM_REPEAT(h, PSM_K_MLP) {
    const float* cur = w.d_xin; int ld_cur = h->ld_in;
    l_first = 0;
    if (c1) {                                  // conv1D_PCA head: Conv1D layers over the scaled coefficients, then Flatten
      int64_t stride = h->ld_in;
      const int nc = (int)h->conv1d.size();
      for (int q = 0; q < nc; ++q) {
        const Conv1dLayer& c = h->conv1d[q];
        PsmConv1dArgs ca{};
        ca.in = cur; ca.in_stride = stride; ca.W = c.W; ca.bias = c.b; ca.out = w.d_c1[q & 1];
        ca.out_stride = q == nc - 1 ? (int64_t)round_up(h->cfg.p_in * c.cout, 32) : (int64_t)h->cfg.p_in * c.cout;
        ca.M = M; ca.P = h->cfg.p_in; ca.k = c.k; ca.c_in = c.cin; ca.c_out = c.cout; ca.relu = 1;
        HIPCHK(h, psm_launch_conv1d(ca, st));
        cur = ca.out; stride = ca.out_stride;
      }
      ld_cur = (int)stride;
    }
    // LayerNormalization (+ residual with the layer's own input) behind a hidden layer: densePCA_attention.  Where the consumer
    // is another hidden Dense launch the normalisation is DEFERRED into it (psm_dense_kernel<..., LNIN>: moments of its own input
    // rows in the prologue, the residual of NNs.py:64 in its epilogue) -- no launch; the last one, whose consumers are the head,
    // the strip-dot riders and the introspection entries, finishes its activation with psm_layernorm_kernel.  PSM_LN_FUSE=0
    // launches every normalisation on its own.
    const char* ln_env = getenv("PSM_LN_FUSE");            // read per solve (diagnostic; the tests switch it in-process)
    const bool ln_fuse = !(ln_env && atoi(ln_env) == 0);
    bool pending = false;                               // `cur` is a raw output whose LayerNormalization the next launch applies
    int pending_l = -1;
    auto after_dense = [&](int l, float* act, const float* layer_in, int ld_layer_in, bool residual_done) -> int {
      const DenseLayer& d = h->dense[l];
      pending = false;
      if (!d.ln) return PSM_OK;
      if (ln_fuse && l + 1 <= nl - 2) { pending = true; pending_l = l; return PSM_OK; }      // the next hidden layer applies it
      PsmLayerNormArgs la{act, d.ldw, (d.ln_residual && !residual_done) ? layer_in : nullptr, ld_layer_in, d.ln_gamma, d.ln_beta, Mpad, d.n_out, d.ln_eps};
      HIPCHK(h, psm_launch_layernorm(la, st));
      return PSM_OK;
    };
    if (fuse1) {
      PsmDenseArgs d0 = dense_args(0, cur, ld_cur);
      HIPCHK(h, psm_launch_reduce_dense1(ra, d0, st));
      int rc0 = after_dense(0, d0.out, cur, ld_cur, false);
      if (rc0) return rc0;
      cur = d0.out; ld_cur = h->dense[0].ldw;
      l_first = 1;
    }
    for (int l = l_first; l < nl; ++l) {
      PsmDenseArgs da = dense_args(l, cur, ld_cur);
      bool residual_done = false;
      if (pending) {                                    // this launch normalises its input (and adds the residual of its own LN)
        const DenseLayer& p = h->dense[pending_l];
        da.ln_gamma = p.ln_gamma; da.ln_beta = p.ln_beta; da.ln_eps = p.ln_eps; da.ln_n = p.n_out;
        da.ln_residual = (h->dense[l].ln && h->dense[l].ln_residual) ? 1 : 0;
        residual_done = da.ln_residual != 0;
      }
      if (l < nl - 1) {
        HIPCHK(h, psm_launch_dense(da, st));
        int rcl = after_dense(l, da.out, cur, ld_cur, residual_done);
        if (rcl) return rcl;
        cur = da.out; ld_cur = h->dense[l].ldw;
        continue;
      }
      if (use_bound && !bf16 && l == nl - 1) { // head layer + strip dots of the bound geometry in one launch
        PsmDotsArgs dd = use_cf ? PsmDotsArgs{h->d_g2p, h->d_c2p, h->d_cntp, h->d_row_of_p, d_row_scale, w.d_dots2, n_cases * CB, h->dense[nl - 1].Kpad, ga, h->B, CB}
                                : PsmDotsArgs{h->d_g2, h->d_c2, h->d_cnt, h->d_row_of, d_row_scale, w.d_dots, h->bound_rows * n_cases, h->dense[nl - 1].Kpad, ga};
        HIPCHK(h, psm_launch_dense_dots(da, dd, st));
      } else {
        HIPCHK(h, psm_launch_dense(da, st));
      }
      cur = da.out; ld_cur = h->dense[l].ldw;
    }
  }
  tm.after(PSM_K_MLP);
  if (use_bound) {
    PsmDecodeArgs de{};
    de.res = w.d_res; de.ld_res = h->ld_out; de.bpack = h->d_bpack_out; de.mean = h->d_mean_out;
    de.row_scale = d_row_scale; de.pred = nullptr; de.M = M; de.Mpad = Mpad; de.Gd = h->Gd;
    de.n_coltiles = h->n_coltiles; de.K_out = h->K_out;
    // decode + paste on a bound geometry: x6 arithmetic by default (single case 8.44 -> 8.16 us, 8 cases 10.2 -> 8.8 us; same
    // accuracy as the float32 MFMA, tools/x6_check.py); PSM_X6 bit 1 = 0 keeps v_mfma_f32_32x32x2_f32
    de.x6 = h->x6_mode < 0 ? 1 : ((h->x6_mode & 2) ? 1 : 0);
    PsmBoundArgs ba{};
    ba.cp = h->plan.cp; ba.blocks = h->d_blocks; ba.dots = w.d_dots; ba.scnt = h->d_cnt; ba.ownbits = h->d_ownbits;
    ba.blk_y0x0 = h->d_blk; ba.shiftW = h->d_shiftW;
    for (int f = 0; f < 2; ++f) ba.shiftL[f] = (int)h->plan.shiftA[f].size();
    ba.fields = d_fields; ba.offs = w.d_offs; ba.shift = w.d_shift; ba.Nx = h->Nx; ba.n_strips = h->n_strips; ba.B = h->B;
    ba.gflags = gflags; ba.n_gwaves = n_gwaves;
    ba.cf = use_cf ? 1 : 0; ba.cf_dots = w.d_dots2; ba.cf_a0 = h->d_cfa0;
    if (h->bound_zero_fill)                    // cells no block covers stay 0 like the reference's np.zeros field
      HIPCHK(h, hipMemsetAsync(d_fields, 0, (size_t)n_cases * h->Ny * h->Nx * h->cfg.c_out * sizeof(float), st));
    if (n_cases == 1 && h->B <= 64) {
      tm.before(PSM_K_DECODE);
      if (bf16) {                               // dots from the bf16-rounded res (own small launch)
        PsmDotsArgs dd = use_cf ? PsmDotsArgs{h->d_g2p, h->d_c2p, h->d_cntp, h->d_row_of_p, d_row_scale, w.d_dots2, CB, h->ld_out, ga, h->B, CB}
                                : PsmDotsArgs{h->d_g2, h->d_c2, h->d_cnt, h->d_row_of, d_row_scale, w.d_dots, h->bound_rows, h->ld_out, ga};
        HIPCHK(h, psm_launch_res_dots(dd, w.d_res, h->ld_out, st));
      }
      PSM_REPEAT(h, PSM_K_DECODE) HIPCHK(h, psm_launch_decode_paste(de, ba, h->cfg.c_out, st, bf16 ? 1 : 0));
      tm.after(PSM_K_DECODE);
      tm.before(PSM_K_STRIPS); tm.after(PSM_K_STRIPS);
      tm.before(PSM_K_CHAIN); tm.after(PSM_K_CHAIN);
      tm.before(PSM_K_PASTE); tm.after(PSM_K_PASTE);
      return PSM_OK;
    }
    // case batch: the chains of all cases in one small launch, then decode + paste over all block rows
    PsmBoundBatchArgs bb{};
    bb.cp = h->plan.cp; bb.blocks = h->d_blocks; bb.dots = w.d_dots; bb.scnt = h->d_cnt; bb.ownbits = h->d_ownbits;
    bb.blk_y0x0 = h->d_blk; bb.shiftW = h->d_shiftW;
    for (int f = 0; f < 2; ++f) bb.shiftL[f] = (int)h->plan.shiftA[f].size();
    bb.fields = d_fields; bb.offs = w.d_offs; bb.shift = w.d_shift; bb.Nx = h->Nx; bb.npix = h->Ny * h->Nx;
    bb.n_strips = h->n_strips; bb.B = h->B; bb.rows_pc = h->bound_rows; bb.n_cases = n_cases;
    bb.gflags = gflags; bb.n_gwaves = n_gwaves;
    bb.cf = use_cf ? 1 : 0; bb.cf_dots = w.d_dots2; bb.cf_a0 = h->d_cfa0;
    tm.before(PSM_K_DECODE); tm.after(PSM_K_DECODE);
    tm.before(PSM_K_STRIPS); tm.after(PSM_K_STRIPS);
    tm.before(PSM_K_CHAIN);
    if (bf16) {                                 // dots from the bf16-rounded res (own small launch): pair rows, or the strip rows of the chain
      PsmDotsArgs dd = use_cf ? PsmDotsArgs{h->d_g2p, h->d_c2p, h->d_cntp, h->d_row_of_p, d_row_scale, w.d_dots2, n_cases * CB, h->ld_out, ga, h->B, CB}
                              : PsmDotsArgs{h->d_g2, h->d_c2, h->d_cnt, h->d_row_of, d_row_scale, w.d_dots, h->bound_rows * n_cases, h->ld_out, ga};
      HIPCHK(h, psm_launch_res_dots(dd, w.d_res, h->ld_out, st));
    }
    if (!use_cf) HIPCHK(h, psm_launch_chain_dots(bb, h->cfg.c_out, st));     // closed form: no chain launch
    tm.after(PSM_K_CHAIN);
    tm.before(PSM_K_PASTE);
    HIPCHK(h, psm_launch_decode_paste_batch(de, bb, h->cfg.c_out, st, bf16 ? 1 : 0));
    tm.after(PSM_K_PASTE);
    return PSM_OK;
  }

  PsmDecodeArgs de{};
  de.res = w.d_res; de.ld_res = h->ld_out; de.bpack = h->d_bpack_out; de.mean = h->d_mean_out;
  de.row_scale = d_row_scale; de.pred = w.d_pred; de.M = M; de.Mpad = Mpad; de.Gd = h->Gd;
  de.n_coltiles = h->n_coltiles; de.K_out = h->K_out;
  tm.before(PSM_K_DECODE);
  PSM_REPEAT(h, PSM_K_DECODE) HIPCHK(h, bf16 ? psm_launch_decode_bf16(de, st) : psm_launch_decode(de, st));
  tm.after(PSM_K_DECODE);

  PsmStripArgs sa{};
  sa.pred = w.d_pred; sa.grid = d_grid; sa.strips = h->d_strips; sa.blk_y0x0 = h->d_blk; sa.spart = w.d_spart; sa.colpart = w.d_colpart; sa.NS = h->plan.cp.NS; sa.n_bands = h->n_bands;
  sa.B = h->B; sa.S = h->S; sa.c_in = h->cfg.c_in; sa.c_out = h->cfg.c_out;
  sa.sdf_ch = h->cfg.sdf_channel; sa.Ny = h->Ny; sa.Nx = h->Nx;
  tm.before(PSM_K_STRIPS);
  PSM_REPEAT(h, PSM_K_STRIPS) HIPCHK(h, psm_launch_strips(sa, n_cases, st));
  tm.after(PSM_K_STRIPS);

  PsmChainArgs ca{};
  ca.cp = h->plan.cp; ca.blocks = h->d_blocks; ca.spart = w.d_spart; ca.colpart = w.d_colpart; ca.n_bands = h->n_bands; ca.pred = w.d_pred; ca.owner = h->d_owner;
  ca.shiftA = h->d_shiftA; ca.shiftB = h->d_shiftB; ca.shiftOwnA = h->d_shiftOwnA; ca.shiftOwnB = h->d_shiftOwnB; ca.shiftW = h->d_shiftW;
  for (int f = 0; f < 2; ++f) ca.shiftL[f] = (int)h->plan.shiftA[f].size();
  ca.Lmax = h->Lmax; ca.offs = w.d_offs; ca.shift = w.d_shift; ca.n_strips = h->n_strips; ca.c_out = h->cfg.c_out; ca.stamps = h->d_stamps;
  PsmPasteArgs pa{w.d_pred, h->d_owner, w.d_offs, w.d_shift, d_fields, h->B, h->S, h->cfg.c_out, h->Ny * h->Nx};
  if (h->fused_assemble && n_cases < 4) {   // few blocks, few cases: every paste workgroup re-runs the chain (one launch
                                            // less); for case batches one chain workgroup per case + a streaming paste
    tm.before(PSM_K_CHAIN);
    tm.after(PSM_K_CHAIN);
    tm.before(PSM_K_PASTE);
    PSM_REPEAT(h, PSM_K_PASTE) HIPCHK(h, psm_launch_assemble(ca, pa, n_cases, st));
    tm.after(PSM_K_PASTE);
    return PSM_OK;
  }
  tm.before(PSM_K_CHAIN);
  HIPCHK(h, psm_launch_chain(ca, n_cases, st));
  tm.after(PSM_K_CHAIN);
  tm.before(PSM_K_PASTE);
  HIPCHK(h, psm_launch_paste(pa, n_cases, st));
  tm.after(PSM_K_PASTE);
  return PSM_OK;
}

int prepare_scale(psm_handle* h, Workspace& w, const float* out_scale, int n_cases, hipStream_t st, const float** d_scale) {
  if (!out_scale) { *d_scale = h->d_ones; return PSM_OK; }
  const int M = n_cases * h->B;
  const int slot = h->scale_pos;
  h->scale_pos = (h->scale_pos + 1) % psm_handle::RING;
  HIPCHK(h, hipEventSynchronize(h->scale_ev[slot]));
  for (int c = 0; c < n_cases; ++c)
    for (int b = 0; b < h->B; ++b) h->h_scale[slot][c * h->B + b] = out_scale[c];
  HIPCHK(h, hipMemcpyAsync(w.d_row_scale, h->h_scale[slot], (size_t)M * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipEventRecord(h->scale_ev[slot], st));
  *d_scale = w.d_row_scale;
  return PSM_OK;
}

int solve_device(psm_handle* h, const float* d_grid, int n_cases, const float* out_scale, float* d_fields,
                 hipStream_t st, hipEvent_t* prof) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!d_grid || !d_fields) return fail(h, PSM_ERR_ARG, "null buffer");
  if (n_cases < 1 || n_cases > h->cfg.max_cases) return fail(h, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!st) st = h->stream;
  const float* d_scale = nullptr;
  int rc = ensure_encode_aux(h, n_cases);
  if (rc) return rc;
  rc = prepare_scale(h, h->ws0, out_scale, n_cases, st, &d_scale);
  if (rc) return rc;
  h->last_cases = n_cases;
  const bool eager = prof || h->timed_kernel >= 0 || !h->use_graph;
  if (eager) return launch_all(h, h->ws0, d_grid, n_cases, d_fields, d_scale, st, prof);
  GraphKey key{(n_cases * 2 + (out_scale ? 1 : 0)) * 2 + ((h->bound && (h->bound_scope == 2 || h->in_mesh_solve)) ? 1 : 0), d_grid, d_fields};
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    if (h->graphs.size() > 64) destroy_graphs(h);
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
    rc = launch_all(h, h->ws0, d_grid, n_cases, d_fields, d_scale, h->stream, nullptr);
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    it = h->graphs.emplace(key, exec).first;
  }
  HIPCHK(h, hipGraphLaunch(it->second, st));
  return PSM_OK;
}

#ifndef PSM_MESH_STAGE_MAX_DEFAULT
// cells up to which psm_solve reads registered input with the stage kernel; above, the DMA engine's higher large-copy rate (49 against
// 40 GB/s over this PCIe link) wins: measured crossover between 44 k (stage 107 / DMA 110 us) and 69 k cells (147 / 142 us),
// profiles/r04_psm_solve.txt
#define PSM_MESH_STAGE_MAX_DEFAULT 50000
#endif
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

// ---- guard of the bound-geometry contract, host side -------------------------------------------------------------
// true once per trip: a guard wave of a solve on workspace `w` found a grid whose flow-cell pattern is not the bound one
bool guard_take(psm_handle* h, Workspace& w) {
  if (!h->h_guard) return false;
  volatile int* f = h->h_guard + w.gidx;
  if (!*f) return false;
  *f = 0;
  return true;
}
// drop the binding: the following solves (and the re-run of the one that tripped) take the general path
int guard_drop(psm_handle* h, const char* where) {
  ++h->guard_trips;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  destroy_graphs(h);
  h->bound = false;
  h->err = std::string(where) + ": the grid's flow-cell pattern (SDF channel != 0) is not the one bound with psm_bind_geometry; the binding was dropped";
  return PSM_OK;
}

}  // namespace

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

// Closed form of the offset chain for a bound case batch.  On a bound geometry every branch of the chain (np.isnan tests,
// the 0.9 coverage test, the first non-empty column) is decided by the strip COUNTS, so the correction of block b plus the
// global shift is a fixed linear map of the strip means: read off the host replay of the chain (psm_chain, double) by
// probing it with unit vectors, checked against a random probe, and folded into one table row per (field, block, source
// block) -- a linear combination of the strip / shift rows the bind kernels have just built.  Leaves bound_cf false (the
// chain launch stays) if the probe disagrees.
static int build_closed_form(psm_handle* h, int n_cases, int rows, int Kh) {
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
static int bind_geometry_device(psm_handle* h, const float* d_grid, int n_cases = 1) {
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

int psm_solve_grid_device(psm_handle* h, const float* d_grid, int32_t n_cases, const float* out_scale,
                          float* d_fields, void* stream) {
  return solve_device(h, d_grid, n_cases, out_scale, d_fields, (hipStream_t)stream, nullptr);
}

static bool host_registered(const psm_handle* h, const void* p, size_t bytes);
int psm_solve_grid(psm_handle* h, const float* grid, int32_t n_cases, const float* out_scale, float* fields) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!grid || !fields) return fail(h, PSM_ERR_ARG, "null buffer");
  if (n_cases < 1 || n_cases > h->cfg.max_cases) return fail(h, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gin = (size_t)n_cases * npix * h->cfg.c_in * sizeof(float);
  const size_t gout = (size_t)n_cases * npix * h->cfg.c_out * sizeof(float);
  const bool reg_in = host_registered(h, grid, gin), reg_out = host_registered(h, fields, gout);
  if (!reg_in) memcpy(h->h_grid, grid, gin);
  HIPCHK(h, hipMemcpyAsync(h->d_grid_stage, reg_in ? grid : h->h_grid, gin, hipMemcpyHostToDevice, h->stream));
  int rc = solve_device(h, h->d_grid_stage, n_cases, out_scale, h->d_fields_stage, h->stream, nullptr);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(reg_out ? fields : h->h_fields, h->d_fields_stage, gout, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, wait_stream(h->stream));
  if (guard_take(h, h->ws0)) {                 // not the bound geometry: the field is NaN -- drop the binding, solve again on the general path
    if ((rc = guard_drop(h, "psm_solve_grid"))) return rc;
    const std::string note = h->err;
    if ((rc = solve_device(h, h->d_grid_stage, n_cases, out_scale, h->d_fields_stage, h->stream, nullptr))) return rc;
    HIPCHK(h, hipMemcpyAsync(reg_out ? fields : h->h_fields, h->d_fields_stage, gout, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    h->err = note + " (solved on the general path)";
  }
  if (!reg_out) memcpy(fields, h->h_fields, gout);
  return PSM_OK;
}

// ---- host-buffer ring ------------------------------------------------------------------------------------------
// Every slot owns pinned host buffers, device buffers, a workspace and a stream; the H2D copy, the kernels and the D2H
// copy of one ticket are ONE hipGraph replay on that stream (one host call per solve), and the slots overlap freely:
// the copies of ticket k+1 / k-1 run on the DMA engines while the kernels of ticket k compute.
static bool host_registered(const psm_handle* h, const void* p, size_t bytes) {
  const char* c = (const char*)p;
  for (auto& r : h->host_regs) if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
  return false;
}
// device-side address of a host pointer inside a registered range (nullptr: not registered / not mapped)
static float* host_mapped(const psm_handle* h, const void* p, size_t bytes) {
  const char* c = (const char*)p;
  for (auto& r : h->host_regs)
    if (c >= r.base && c + bytes <= r.base + r.bytes) return r.dev ? (float*)(r.dev + (c - r.base)) : nullptr;
  return nullptr;
}

static int ring_init(psm_handle* h) {
  if (h->ring_ready) return PSM_OK;
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gin = (size_t)h->cfg.max_cases * npix * h->cfg.c_in, gout = (size_t)h->cfg.max_cases * npix * h->cfg.c_out;
  const char* rg = getenv("PSM_RING_GRAPH");
  h->ring_graph = (rg && rg[0] == '0') ? 0 : 1;
  const char* ru = getenv("PSM_RING_USE");
  h->ring_slots = (ru && atoi(ru) >= 1 && atoi(ru) <= psm_handle::SLOTS) ? atoi(ru) : psm_handle::SLOTS;
  const char* rp = getenv("PSM_RING_PULL");
  h->ring_dma = (rp && rp[0] == '1') ? 0 : 1;
  for (auto& s : h->slot) {
    HIPCHK(h, hipHostMalloc((void**)&s.h_in, gin * sizeof(float), hipHostMallocMapped));
    HIPCHK(h, hipHostMalloc((void**)&s.h_out, gout * sizeof(float), hipHostMallocMapped));
    HIPCHK(h, hipHostMalloc((void**)&s.h_rs, (size_t)h->Mpad_cap * sizeof(float), hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&s.m_in, s.h_in, 0) != hipSuccess || hipHostGetDevicePointer((void**)&s.m_out, s.h_out, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&s.m_rs, s.h_rs, 0) != hipSuccess) {
      (void)hipGetLastError();
      s.m_in = s.m_out = s.m_rs = nullptr;
      h->ring_dma = 1;                                   // no mapped view of pinned memory: DMA copies
    }
    int rc;
    if ((rc = dev_alloc(h, &s.d_in, gin))) return rc;
    if ((rc = dev_alloc(h, &s.d_out, gout))) return rc;
    if ((rc = ws_alloc(h, s.ws))) return rc;
    HIPCHK(h, hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    s.state = 0; s.ticket = -1;
  }
  HIPCHK(h, hipDeviceSynchronize());
  h->ring_ready = true;
  return PSM_OK;
}

// key of a captured slot graph: everything the captured launch sequence depends on
static int ring_key(const psm_handle* h, int n_cases, bool scale) {
  const bool bound = h->bound && h->bound_scope == 2 && n_cases == h->bound_cases;
  return ((n_cases * 2 + (scale ? 1 : 0)) * 2 + (bound ? 1 : 0)) * 2 + (h->ring_dma ? 1 : 0);
}

// The launch sequence of one ticket on the slot's stream.
//  pull form (src_dev / dst_dev = device-side addresses of pinned or registered host memory): a stage-in kernel pulls the
//    grid over PCIe into s.d_in (and expands the out_scale), the solve's last kernel stores the field straight into
//    dst_dev -- kernels only;
//  DMA form (src_dev == nullptr): hipMemcpyAsync H2D from src, kernels, hipMemcpyAsync D2H into dst (copies optional:
//    with_copies = false enqueues the kernels alone).
static int ring_sequence(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
                         const float* src, float* dst, bool with_copies) {
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t nin = (size_t)n_cases * npix * h->cfg.c_in, nout = (size_t)n_cases * npix * h->cfg.c_out;
  const int M = n_cases * h->B;
  if (src_dev) {
    static const int dbg = getenv("PSM_RING_DEBUG") ? atoi(getenv("PSM_RING_DEBUG")) : 0;   // timing experiments only: 1 no stage-in, 2 field stays on the device
    if (!(dbg & 1)) HIPCHK(h, psm_launch_stage_in(src_dev, s.d_in, nin, scale ? s.m_rs : nullptr, s.ws.d_row_scale, M, h->B, s.st));
    return launch_all(h, s.ws, s.d_in, n_cases, (dbg & 2) ? s.d_out : dst_dev, scale ? s.ws.d_row_scale : h->d_ones, s.st, nullptr);
  }
  if (with_copies) HIPCHK(h, hipMemcpyAsync(s.d_in, src, nin * sizeof(float), hipMemcpyHostToDevice, s.st));
  if (scale) HIPCHK(h, hipMemcpyAsync(s.ws.d_row_scale, s.h_rs, (size_t)M * sizeof(float), hipMemcpyHostToDevice, s.st));
  int rc = launch_all(h, s.ws, s.d_in, n_cases, s.d_out, scale ? s.ws.d_row_scale : h->d_ones, s.st, nullptr);
  if (rc) return rc;
  if (with_copies) HIPCHK(h, hipMemcpyAsync(dst, s.d_out, nout * sizeof(float), hipMemcpyDeviceToHost, s.st));
  return PSM_OK;
}

static int ring_capture(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
                        bool with_copies, hipGraphExec_t* out) {
  hipGraph_t graph = nullptr;
  HIPCHK(h, hipStreamBeginCapture(s.st, hipStreamCaptureModeRelaxed));
  int rc = ring_sequence(h, s, n_cases, scale, src_dev, dst_dev, s.h_in, s.h_out, with_copies);
  hipError_t e2 = hipStreamEndCapture(s.st, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e2 != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(h, PSM_ERR_HIP, std::string("ring capture: ") + hipGetErrorString(e2));
  }
  hipError_t e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(h, PSM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  return PSM_OK;
}

// Enqueue one ticket.  src / dst: where the grid is read from / the field is written to (the slot's pinned buffers or
// registered caller memory).
static int ring_launch(psm_handle* h, psm_handle::Slot& s, int n_cases, const float* out_scale, const float* src, float* dst) {
  { int rc0 = ensure_encode_aux(h, n_cases); if (rc0) return rc0; }
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gin = (size_t)n_cases * npix * h->cfg.c_in * sizeof(float), gout = (size_t)n_cases * npix * h->cfg.c_out * sizeof(float);
  const bool scale = out_scale != nullptr;
  const bool own = (src == s.h_in && dst == s.h_out);
  if (h->h_guard) h->h_guard[s.ws.gidx] = 0;            // the slot is free: nothing of an earlier ticket can still raise it
  s.last_src = src; s.last_dst = dst;
  if (scale) { if (out_scale != s.last_scale.data()) s.last_scale.assign(out_scale, out_scale + n_cases); } else s.last_scale.clear();
  const float* src_dev = nullptr;
  float* dst_dev = nullptr;
  if (!h->ring_dma) {                                    // pull form needs device-side views of both host buffers
    src_dev = src == s.h_in ? s.m_in : host_mapped(h, src, gin);
    dst_dev = dst == s.h_out ? s.m_out : host_mapped(h, dst, gout);
    if (!src_dev || !dst_dev) src_dev = nullptr, dst_dev = nullptr;
  }
  if (scale) {
    if (src_dev) for (int c = 0; c < n_cases; ++c) s.h_rs[c] = out_scale[c];
    else
      for (int c = 0; c < n_cases; ++c)
        for (int b = 0; b < h->B; ++b) s.h_rs[c * h->B + b] = out_scale[c];
  }
  const int key = ring_key(h, n_cases, scale);
  const bool graphs = h->ring_graph && h->timed_kernel < 0;
  int rc;
  if (src_dev && graphs && own) {                        // pull form on the slot's own buffers: the whole ticket is one replay
    if (!s.g_full || s.g_full_key != key) {
      if (s.g_full) { (void)hipGraphExecDestroy(s.g_full); s.g_full = nullptr; }
      if ((rc = ring_capture(h, s, n_cases, scale, src_dev, dst_dev, true, &s.g_full))) return rc;
      s.g_full_key = key;
    }
    HIPCHK(h, hipGraphLaunch(s.g_full, s.st));
  } else if (src_dev || !graphs) {                       // pull form on caller memory (pointers differ per ticket) / plain launches
    if ((rc = ring_sequence(h, s, n_cases, scale, src_dev, dst_dev, src, dst, true))) return rc;
  } else {
    // Default: the two copies are hipMemcpyAsync calls on the slot's stream (DMA engines; inside a graph they would
    // become blit kernels, which read host memory at ~20 GB/s), the kernels in between are one graph replay.
    static const int dbg = getenv("PSM_RING_DEBUG") ? atoi(getenv("PSM_RING_DEBUG")) : 0;   // timing experiments only: 1 no H2D, 2 no D2H
    if (!(dbg & 1)) HIPCHK(h, hipMemcpyAsync(s.d_in, src, gin, hipMemcpyHostToDevice, s.st));
    if (!s.g_kern || s.g_kern_key != key) {
      if (s.g_kern) { (void)hipGraphExecDestroy(s.g_kern); s.g_kern = nullptr; }
      if ((rc = ring_capture(h, s, n_cases, scale, nullptr, nullptr, false, &s.g_kern))) return rc;
      s.g_kern_key = key;
    }
    HIPCHK(h, hipGraphLaunch(s.g_kern, s.st));
    if (!(dbg & 2)) HIPCHK(h, hipMemcpyAsync(dst, s.d_out, gout, hipMemcpyDeviceToHost, s.st));
  }
  HIPCHK(h, hipEventRecord(s.ev_out, s.st));
  return PSM_OK;
}

static int ring_check(psm_handle* h, int32_t n_cases) {
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (n_cases < 1 || n_cases > h->cfg.max_cases) return fail(h, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  return PSM_OK;
}

// A ticket whose grid was not the bound geometry (its field is NaN): drop the binding and run the ticket again on the
// general path, from the same source into the same destination.
static int ring_guard_rerun(psm_handle* h, psm_handle::Slot& s, const char* where) {
  if (!guard_take(h, s.ws)) return PSM_OK;
  int rc = guard_drop(h, where);
  if (rc) return rc;
  const std::string note = h->err;
  std::vector<float> sc = s.last_scale;
  if ((rc = ring_launch(h, s, s.n_cases, sc.empty() ? nullptr : sc.data(), s.last_src, s.last_dst))) return rc;
  HIPCHK(h, wait_event(s.ev_out));
  h->err = note + " (ticket solved again on the general path)";
  return PSM_OK;
}

int psm_ring_acquire(psm_handle* h, int64_t* ticket, float** grid_in, float** fields_out) {
  if (!h) return PSM_ERR_ARG;
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (!ticket || !grid_in || !fields_out) return fail(h, PSM_ERR_ARG, "null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  int rc = ring_init(h);
  if (rc) return rc;
  psm_handle::Slot& s = h->slot[h->next_ticket % h->ring_slots];
  if (s.state != 0)
    return fail(h, PSM_ERR_STATE, "submission ring full: wait for the oldest ticket first (PSM_RING_SLOTS in flight)");
  s.state = 1; s.ticket = h->next_ticket; s.user_out = nullptr; s.direct_out = false;
  *ticket = h->next_ticket++;
  *grid_in = s.h_in; *fields_out = s.h_out;
  return PSM_OK;
}

static int slot_of(psm_handle* h, int64_t ticket, int state, psm_handle::Slot** out);
int psm_ring_release(psm_handle* h, int64_t ticket) {
  if (!h) return PSM_ERR_ARG;
  psm_handle::Slot* s = nullptr;
  int rc = slot_of(h, ticket, 1, &s);                  // acquired, not submitted
  if (rc) return rc;
  s->state = 0;
  // the slot comes round again PSM_RING_SLOTS tickets later; the ticket counter does not go back (tickets stay unique)
  return PSM_OK;
}

static int slot_of(psm_handle* h, int64_t ticket, int state, psm_handle::Slot** out) {
  if (ticket < 0 || !h->ring_ready) return fail(h, PSM_ERR_ARG, "unknown ticket");
  psm_handle::Slot& s = h->slot[ticket % h->ring_slots];
  if (s.ticket != ticket || s.state != state)
    return fail(h, PSM_ERR_ARG, state == 1 ? "unknown ticket (not acquired, or already submitted)" : "unknown ticket (never submitted or already waited for)");
  *out = &s;
  return PSM_OK;
}

int psm_ring_submit(psm_handle* h, int64_t ticket, int32_t n_cases, const float* out_scale) {
  if (!h) return PSM_ERR_ARG;
  int rc = ring_check(h, n_cases);
  if (rc) return rc;
  psm_handle::Slot* s = nullptr;
  if ((rc = slot_of(h, ticket, 1, &s))) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if ((rc = ring_launch(h, *s, n_cases, out_scale, s->h_in, s->h_out))) return rc;
  s->state = 2; s->n_cases = n_cases;
  return PSM_OK;
}

int psm_ring_wait(psm_handle* h, int64_t ticket) {
  if (!h) return PSM_ERR_ARG;
  psm_handle::Slot* s = nullptr;
  int rc = slot_of(h, ticket, 2, &s);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, wait_event(s->ev_out));
  if ((rc = ring_guard_rerun(h, *s, "psm_ring_wait"))) return rc;
  s->state = 0;
  return PSM_OK;
}

int psm_submit_grid_io(psm_handle* h, const float* grid, int32_t n_cases, const float* out_scale, float* fields, int64_t* ticket) {
  if (!h) return PSM_ERR_ARG;
  int rc = ring_check(h, n_cases);
  if (rc) return rc;
  if (!grid || !ticket) return fail(h, PSM_ERR_ARG, "null argument");
  int64_t t; float *gi, *fo;
  if ((rc = psm_ring_acquire(h, &t, &gi, &fo))) return rc;
  psm_handle::Slot& s = h->slot[t % h->ring_slots];
  const size_t npix = (size_t)h->Ny * h->Nx;
  const size_t gin = (size_t)n_cases * npix * h->cfg.c_in * sizeof(float), gout = (size_t)n_cases * npix * h->cfg.c_out * sizeof(float);
  const float* src = s.h_in;
  float* dst = s.h_out;
  if (host_registered(h, grid, gin)) src = grid;                       // DMA straight from the caller's memory
  else memcpy(s.h_in, grid, gin);                                      // caller's buffer is free on return
  if (fields && host_registered(h, fields, gout)) { dst = fields; s.direct_out = true; }
  s.user_out = fields;
  if ((rc = ring_launch(h, s, n_cases, out_scale, src, dst))) { s.state = 0; return rc; }
  s.state = 2; s.n_cases = n_cases;
  *ticket = t;
  return PSM_OK;
}

int psm_submit_grid(psm_handle* h, const float* grid, int32_t n_cases, const float* out_scale, int64_t* ticket) {
  return psm_submit_grid_io(h, grid, n_cases, out_scale, nullptr, ticket);
}

int psm_wait_grid(psm_handle* h, int64_t ticket, float* fields) {
  if (!h) return PSM_ERR_ARG;
  psm_handle::Slot* s = nullptr;
  int rc = slot_of(h, ticket, 2, &s);
  if (rc) return rc;
  if (!fields) fields = s->user_out;
  if (!fields) return fail(h, PSM_ERR_ARG, "null buffer (no destination was given at submission either)");
  if (s->direct_out && fields != s->user_out) return fail(h, PSM_ERR_ARG, "this ticket's field was DMA'd into the buffer given at submission");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, wait_event(s->ev_out));
  if ((rc = ring_guard_rerun(h, *s, "psm_wait_grid"))) return rc;
  if (!s->direct_out) memcpy(fields, s->h_out, (size_t)s->n_cases * h->Ny * h->Nx * h->cfg.c_out * sizeof(float));
  s->state = 0;
  return PSM_OK;
}

int psm_host_register(psm_handle* h, void* ptr, size_t bytes) {
  if (!h) return PSM_ERR_ARG;
  if (!ptr || bytes == 0) return fail(h, PSM_ERR_ARG, "null range");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  for (auto& r : h->host_regs) if (r.base == (char*)ptr) return fail(h, PSM_ERR_STATE, "range already registered");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, PSM_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }   // DMA copies only
  h->host_regs.push_back({(char*)ptr, bytes, (char*)dev});
  return PSM_OK;
}

int psm_host_unregister(psm_handle* h, void* ptr) {
  if (!h) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  for (size_t i = 0; i < h->host_regs.size(); ++i)
    if (h->host_regs[i].base == (char*)ptr) {
      HIPCHK(h, hipDeviceSynchronize());                  // no DMA of this handle may still touch the range
      (void)hipHostUnregister(ptr);
      h->host_regs.erase(h->host_regs.begin() + i);
      return PSM_OK;
    }
  return fail(h, PSM_ERR_ARG, "range was not registered with this handle");
}

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

int psm_synchronize(psm_handle* h) {
  if (!h) return PSM_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipDeviceSynchronize());
  if (guard_take(h, h->ws0)) {                 // a psm_solve_grid_device call on another geometry: its field is NaN
    int rc = guard_drop(h, "psm_solve_grid_device");
    return rc ? rc : PSM_ERR_GEOMETRY;
  }
  return PSM_OK;
}

int64_t psm_guard_trips(const psm_handle* h) { return h ? h->guard_trips : -1; }

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
        const float* act = bf ? h->ws0.d_res : h->ws0.d_act[(nl - 2) & 1];
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

// Dispatch-level time of EVERY kernel of the solve path: `steps` solves through the same launch sequence as
// psm_solve_grid_device, each dispatch stamped by hipExtLaunchKernelGGL (its own begin / end, what rocprofv3 reads).
// every dispatch of `steps` solves with its own begin / end stamps: per kernel (launch order of first appearance) the samples in ms
static int collect_kernel_samples(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps,
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
