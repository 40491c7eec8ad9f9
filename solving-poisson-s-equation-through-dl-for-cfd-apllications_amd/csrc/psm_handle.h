// psm_handle.h -- INTERNAL to libpsm_hip.so: the handle behind include/psm.h and the helpers its translation units share.
// The C-ABI is implemented in five files along the seams of the path (nothing here is exported: namespace psm_impl is hidden):
//   psm_api_model.cpp       psm_create / psm_destroy, model artefacts (PCA bases, scaler, Dense / Conv1D / attention / LayerNorm), packing
//   psm_api_plan.cpp        psm_plan_grid (block layout, workspaces), psm_bind_geometry* (bound-geometry tables, closed-form chain)
//   psm_api_solve.cpp       the launch sequence of one solve (launch_all), psm_solve_grid*, the pinned submission ring
//   psm_api_mesh.cpp        the solver boundary (psm_set_geometry / psm_solve*), evaluator helpers (labels, block error, filters, integration)
//   psm_api_introspect.cpp  psm_read_stage, profiling and kernel timing, host-side reference reassembly
// Compiled with hipcc for gfx950 only.  There is no CPU fallback: without a usable device psm_create fails with PSM_ERR_NO_DEVICE.
#pragma once
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

namespace psm_impl __attribute__((visibility("hidden"))) {
extern thread_local std::string g_create_error;   // message of the last failed psm_create (psm_api_model.cpp)

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
  float* d_act_rows = nullptr;          // row-major copy of the last hidden activation when the chain between the Dense launches is packed (large batches)
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
}  // namespace psm_impl
using namespace psm_impl;

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
  bool last_act_packed = false;         // the last solve on ws0 left its last hidden activation in MFMA operand order (PsmDenseArgs::out_packed)
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

namespace psm_impl __attribute__((visibility("hidden"))) {

// ---- helpers shared by the translation units (defined in the file named in the list above) ----
int local_ranks_from_env();
bool sync_blocks();
hipError_t wait_stream(hipStream_t st);
hipError_t wait_event(hipEvent_t ev);
int fail(psm_handle* h, int code, const std::string& msg);
int scratch_reserve(psm_handle* h, size_t dev_bytes, size_t pin_bytes);
void destroy_graphs(psm_handle* h);
void ws_free(Workspace& w);
int ws_alloc_guard(psm_handle* h, Workspace& w);
int ws_alloc(psm_handle* h, Workspace& w);
void ring_drop_graphs(psm_handle* h);
void free_plan(psm_handle* h);
std::vector<float4> pack_comp_in(const double* comp, int P, int K, int c_in, int S, int NT);
std::vector<float4> pack_comp_out(const double* comp, int P, int K_out, int Gd);
void unpin_buffers(psm_handle* h);
void free_geometry(psm_handle* h);
std::vector<uint16_t> pack_comp_in_bf16(const double* comp, int P, int K, int c_in, int S, int NT);
std::vector<uint16_t> pack_comp_out_bf16(const double* comp, int P, int K_out, int G);
bool model_complete(const psm_handle* h);
int encode_groups(const psm_handle* h, int Mpad);
int ensure_encode_aux(psm_handle* h, int n_cases);
int launch_all(psm_handle* h, Workspace& w, const float* d_grid, int n_cases, float* d_fields, const float* d_row_scale,
               hipStream_t st, hipEvent_t* prof);
int prepare_scale(psm_handle* h, Workspace& w, const float* out_scale, int n_cases, hipStream_t st, const float** d_scale);
int solve_device(psm_handle* h, const float* d_grid, int n_cases, const float* out_scale, float* d_fields,
                 hipStream_t st, hipEvent_t* prof);
int mesh_sequence(psm_handle* h, int64_t n, hipStream_t st);
bool guard_take(psm_handle* h, Workspace& w);
int guard_drop(psm_handle* h, const char* where);
int build_closed_form(psm_handle* h, int n_cases, int rows, int Kh);
int bind_geometry_device(psm_handle* h, const float* d_grid, int n_cases = 1);
bool host_registered(const psm_handle* h, const void* p, size_t bytes);
float* host_mapped(const psm_handle* h, const void* p, size_t bytes);
int ring_init(psm_handle* h);
int ring_key(const psm_handle* h, int n_cases, bool scale);
int ring_sequence(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
                         const float* src, float* dst, bool with_copies);
int ring_capture(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
                        bool with_copies, hipGraphExec_t* out);
int ring_launch(psm_handle* h, psm_handle::Slot& s, int n_cases, const float* out_scale, const float* src, float* dst);
int ring_check(psm_handle* h, int32_t n_cases);
int ring_guard_rerun(psm_handle* h, psm_handle::Slot& s, const char* where);
int slot_of(psm_handle* h, int64_t ticket, int state, psm_handle::Slot** out);
int collect_kernel_samples(psm_handle* h, const float* d_grid, int32_t n_cases, float* d_fields, int32_t steps,
                                  std::vector<std::string>& seen, std::vector<std::vector<float>>& samp);

inline double spin_budget_us() { static const double v = [] { const char* e = getenv("PSM_SPIN_US"); return e ? atof(e) : 300.0; }(); return v; }

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

// carve helpers: 256-byte aligned pieces of the two scratch buffers
struct Carver {
  char* base; size_t off = 0;
  template <typename T> T* take(size_t n) { T* p = reinterpret_cast<T*>(base + off); off += (n * sizeof(T) + 255) & ~(size_t)255; return p; }
};

inline size_t carve_size(std::initializer_list<size_t> bytes) { size_t t = 0; for (size_t b : bytes) t += (b + 255) & ~(size_t)255; return t; }


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


inline uint16_t f2bf(double v) {             // round-to-nearest-even float -> bf16 (NaN stays NaN)
  const float f = (float)v;
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
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


#ifndef PSM_MESH_STAGE_MAX_DEFAULT

// cells up to which psm_solve reads registered input with the stage kernel; above, the DMA engine's higher large-copy rate (49 against
// 40 GB/s over this PCIe link) wins: measured crossover between 44 k (stage 107 / DMA 110 us) and 69 k cells (147 / 142 us),
// profiles/archive/r04_psm_solve.txt
#define PSM_MESH_STAGE_MAX_DEFAULT 50000

#endif

}  // namespace psm_impl
