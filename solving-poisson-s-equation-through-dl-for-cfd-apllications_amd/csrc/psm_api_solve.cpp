// psm_api_solve.cpp -- C-ABI of libpsm_hip.so (include/psm.h): one solve: launch sequence, host-buffer entries, pinned ring.  See psm_handle.h for the map of the five files.
#include "psm_handle.h"

namespace psm_impl {


// K groups of the M-tiled x6 encode (psm_encode_x6_mt_kernel) for Mpad block rows; 1 = the one-slab-per-slice form (psm_encode_x6_kernel).
// A workgroup = one K group x 64 block rows, two workgroups per CU: the group count decides both the slab bytes and how the
// workgroups fill 512 slots.  Measured on one box, whole step of the deltas case batch in us (g1 = one slab per slice):
//   cases  rows(pad)  row groups   g1      256 groups  128     64      uneven 512 / row groups
//     8      96         2          42.2    41.0
//    12     128         2          47.9    44.6
//    16     160         3          59.8    56.1        56.8    59.1
//    20     192         3          64.6    60.0        59.7    61.5
//    28     256         4          74.5    70.1        65.7    65.9
//    32     288         5          88.9    84.1        85.4    86.3    92.1 (102 groups)
//    40     384         6         102.0    95.7        91.4    92.0
//    48     448         7         113.8   106.3       104.0    99.1   110.6 (73)
//    56     512         8         124.7   116.9       110.6   106.5
//    64     576         9         140.7   132.1       130.4   134.9   123.5 (56)
// Even groups (a power of two) by row-group count from that table; from nine row groups up 512 / row groups (uneven by one slice).
// Below 96 rows (4 cases: 37.6 against 38.0 us) the launch is bound by the basis stream and the one-slab-per-slice form stays.
// PSM_ENCODE_KGROUPS=n forces the group count (1: the old form), PSM_ENCODE_MT_MIN_ROWS the first row count.
int encode_groups(const psm_handle* h, int Mpad) {
  static const int min_rows = getenv("PSM_ENCODE_MT_MIN_ROWS") ? atoi(getenv("PSM_ENCODE_MT_MIN_ROWS")) : 96;
  if (h->cfg.precision == PSM_PRECISION_BF16 || h->NT > 4 || Mpad % 32 != 0 || Mpad < min_rows || ((PSM_PIX_PER_SLICE * h->cfg.c_in) % 32) != 0) return 1;
  if (h->x6_mode >= 0 && !(h->x6_mode & 1)) return 1;
  // psm_encode_x6_mt_kernel addresses the grids of the whole case batch with unsigned 32-bit BYTE offsets from a scalar base
  if ((int64_t)h->cfg.max_cases * h->Ny * h->Nx * h->cfg.c_in >= ((int64_t)1 << 30)) return 1;
  static const int kg_env = getenv("PSM_ENCODE_KGROUPS") ? atoi(getenv("PSM_ENCODE_KGROUPS")) : 0;
  const int row_groups = (Mpad + PSM_ENC_MT_ROWS - 1) / PSM_ENC_MT_ROWS;
  static const int by_rg[9] = {0, 256, 256, 256, 128, 256, 128, 64, 64};
  int groups = row_groups <= 8 ? by_rg[row_groups] : std::max(1, 512 / row_groups);
  groups = std::max((h->n_slices + 7) / 8, std::min(h->n_slices, groups));
  if (kg_env > 0) groups = kg_env;
  return (groups > 1 && groups <= h->n_slices && (h->n_slices + groups - 1) / groups <= 8) ? groups : 1;
}
// Slabs the float32 / x6 encode launch (psm_launch_encode, psm_kernels.hip) writes for these arguments = what the reduce behind it must sum.
// ONE place mirrors the launcher's three branches (the count used to be rebuilt step by step inside launch_all).
static int encode_slabs(const PsmEncodeArgs& a) {
  const int n_slices = a.S * a.S / PSM_PIX_PER_SLICE;
  if (a.x6 && a.kgroup > 1) return a.kgroup;              // M-tiled x6 form: one slab per K group
  return psm_encode_pairs(a) ? n_slices / 2 : n_slices;   // two slices per workgroup, or one slab per slice
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
  {
    const int groups = (ea.x6 && !bf16) ? encode_groups(h, Mpad) : 1;
    if (groups > 1 && h->d_bpack_x6) { ea.kgroup = groups; ea.bpack_x6 = h->d_bpack_x6; }
  }
  // a single case of four component tiles: two K slices per workgroup, half the slabs (psm_encode_pair_kernel)
  ea.pairs_ok = bf16 ? 0 : 1;
  const int n_slabs = bf16 ? h->n_slices : encode_slabs(ea);          // bf16 handles (psm_launch_encode_bf16): one slab per slice

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
  const int guard_wgs = (h->guard_waves + PSM_GUARD_WG_WAVES - 1) / PSM_GUARD_WG_WAVES;
  const bool guard = use_bound && h->guard_on && h->bound_scope == 2 && h->d_maskbits && w.d_gflags;
  if (guard) {
    ga.sdf = d_grid + h->cfg.sdf_channel; ga.bits = h->d_maskbits; ga.flags = w.d_gflags;
    ga.host_flag = h->m_guard ? h->m_guard + w.gidx : nullptr;
    ga.npix = (long long)n_cases * h->Ny * h->Nx; ga.c_in = h->cfg.c_in; ga.n_ballots = h->guard_ballots; ga.n_waves = h->guard_waves;
    ga.wg_first = 0; ga.wg_count = guard_wgs;            // all of them behind the head layer, unless dealt out below
  }
  const float* gflags = guard ? w.d_gflags : h->d_gzero;
  const int n_gwaves = guard ? guard_wgs : 1;              // flags the decode launch sums: one per guard workgroup
  // case batches (and single cases of more than 64 blocks) on a bound geometry: closed form of the chain where it was built
  const bool use_cf = use_bound && h->bound_cf && w.d_dots2;
  const int CB = h->cfg.c_out * h->B;
  if (&w == &h->ws0) { h->last_row_scale = d_row_scale; h->last_used_cf = use_cf; }
  PsmReduceArgs ra{w.d_part, w.d_xin, h->d_ia, h->d_ib, n_slabs, Mpad, h->ld_in};
  const int nl = (int)h->dense.size();
  // Large case batches (more than 128 block rows): the activation between two plain float32 Dense launches in MFMA operand order
  // (PsmDenseArgs::in_packed / out_packed): the consumer's row loads are then contiguous KiBs -- 64 cases: 10.0 -> 8.4 us per hidden
  // layer, 32 cases: 6.7 -> 5.2 (profiles/r06_case_batch.txt).  Not with a LayerNormalization behind the producer (its kernel and the
  // fused form read rows), not for bf16 handles, not in front of the Conv1D head.  PSM_DENSE_PACKED=0 keeps rows everywhere.
  static const bool packed_env = !(getenv("PSM_DENSE_PACKED") && atoi(getenv("PSM_DENSE_PACKED")) == 0);
  auto packable = [&](int l) {          // output of layer l
    if (!packed_env || l < 0 || l >= nl - 1 || Mpad <= 128 || bf16) return false;
    for (const DenseLayer& q : h->dense) if (q.ln) return false;       // densePCA_attention: its normalisations (and their residuals) read rows
    const DenseLayer& d = h->dense[l];
    return d.ldw % 16 == 0 && h->dense[l + 1].Kp <= d.ldw;
  };
  if (&w == &h->ws0) h->last_act_packed = packable(nl - 2);           // psm_read_stage then takes the row-major copy (d_act_rows)
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
    da.out_packed = packable(l) ? 1 : 0;                 // hidden activation of a large batch: written in MFMA operand order ...
    da.in_packed = (l > 0 && packable(l - 1)) ? 1 : 0;   // ... and read as such by the next Dense launch
    // the head's strip-dot riders (and psm_read_stage) read whole rows: the last hidden layer also leaves a row-major copy
    if (da.out_packed && l == nl - 2) da.out_rows = w.d_act_rows;
    if (da.in_packed && head) da.in_rows = w.d_act_rows;
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
    // Large case batches: the guard workgroups (one per 4096 pixels: 1024 for 64 cases, 50 MB of grid to look at again) are dealt
    // evenly over the hidden-layer launches and the head launch instead of all riding behind the head (64 cases: 14.9 us for a
    // 4.3 us layer).  PSM_GUARD_SPREAD=0 keeps them behind the head.
    static const bool spread_env = !(getenv("PSM_GUARD_SPREAD") && atoi(getenv("PSM_GUARD_SPREAD")) == 0);
    const int carriers = nl - 1 - (fuse1 ? 1 : 0);       // hidden layers launched through psm_launch_dense
    const bool spread = guard && spread_env && guard_wgs > 256 && carriers > 0;
    const int share = spread ? guard_wgs / (carriers + 1) : 0;
    int dealt = 0;
    auto riders = [&]() -> const PsmGuardArgs* {         // the next hidden launch's share
      if (!spread) return nullptr;
      ga.wg_first = dealt; ga.wg_count = share; dealt += share;
      return &ga;
    };
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
        HIPCHK(h, psm_launch_dense(da, st, riders()));
        int rcl = after_dense(l, da.out, cur, ld_cur, residual_done);
        if (rcl) return rcl;
        cur = da.out; ld_cur = h->dense[l].ldw;
        continue;
      }
      ga.wg_first = dealt; ga.wg_count = guard ? guard_wgs - dealt : 0;    // the head launch (or the bf16 handles' dots launch) carries the rest
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
    {
      static const bool bufst = !(getenv("PSM_PASTE_BUFFER_STORES") && atoi(getenv("PSM_PASTE_BUFFER_STORES")) == 0);
      const int64_t fb = (int64_t)n_cases * h->Ny * h->Nx * h->cfg.c_out * (int64_t)sizeof(float);
      ba.field_bytes = (bufst && fb < ((int64_t)1 << 32)) ? (uint32_t)fb : 0u;
    }
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
    bb.field_bytes = ba.field_bytes;
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


// ---- host-buffer ring ------------------------------------------------------------------------------------------
// Every slot owns pinned host buffers, device buffers, a workspace and a stream; the H2D copy, the kernels and the D2H
// copy of one ticket are ONE hipGraph replay on that stream (one host call per solve), and the slots overlap freely:
// the copies of ticket k+1 / k-1 run on the DMA engines while the kernels of ticket k compute.
bool host_registered(const psm_handle* h, const void* p, size_t bytes) {
  const char* c = (const char*)p;
  for (auto& r : h->host_regs) if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
  return false;
}

// device-side address of a host pointer inside a registered range (nullptr: not registered / not mapped)
float* host_mapped(const psm_handle* h, const void* p, size_t bytes) {
  const char* c = (const char*)p;
  for (auto& r : h->host_regs)
    if (c >= r.base && c + bytes <= r.base + r.bytes) return r.dev ? (float*)(r.dev + (c - r.base)) : nullptr;
  return nullptr;
}


int ring_init(psm_handle* h) {
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
int ring_key(const psm_handle* h, int n_cases, bool scale) {
  const bool bound = h->bound && h->bound_scope == 2 && n_cases == h->bound_cases;
  return ((n_cases * 2 + (scale ? 1 : 0)) * 2 + (bound ? 1 : 0)) * 2 + (h->ring_dma ? 1 : 0);
}


// The launch sequence of one ticket on the slot's stream.
//  pull form (src_dev / dst_dev = device-side addresses of pinned or registered host memory): a stage-in kernel pulls the
//    grid over PCIe into s.d_in (and expands the out_scale), the solve's last kernel stores the field straight into
//    dst_dev -- kernels only;
//  DMA form (src_dev == nullptr): hipMemcpyAsync H2D from src, kernels, hipMemcpyAsync D2H into dst (copies optional:
//    with_copies = false enqueues the kernels alone).
int ring_sequence(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
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


int ring_capture(psm_handle* h, psm_handle::Slot& s, int n_cases, bool scale, const float* src_dev, float* dst_dev,
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
int ring_launch(psm_handle* h, psm_handle::Slot& s, int n_cases, const float* out_scale, const float* src, float* dst) {
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


int ring_check(psm_handle* h, int32_t n_cases) {
  if (!h->planned) return fail(h, PSM_ERR_STATE, "psm_plan_grid has not been called");
  if (n_cases < 1 || n_cases > h->cfg.max_cases) return fail(h, PSM_ERR_ARG, "n_cases outside [1, max_cases]");
  return PSM_OK;
}


// A ticket whose grid was not the bound geometry (its field is NaN): drop the binding and run the ticket again on the
// general path, from the same source into the same destination.
int ring_guard_rerun(psm_handle* h, psm_handle::Slot& s, const char* where) {
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


int slot_of(psm_handle* h, int64_t ticket, int state, psm_handle::Slot** out) {
  if (ticket < 0 || !h->ring_ready) return fail(h, PSM_ERR_ARG, "unknown ticket");
  psm_handle::Slot& s = h->slot[ticket % h->ring_slots];
  if (s.ticket != ticket || s.state != state)
    return fail(h, PSM_ERR_ARG, state == 1 ? "unknown ticket (not acquired, or already submitted)" : "unknown ticket (never submitted or already waited for)");
  *out = &s;
  return PSM_OK;
}

}  // namespace psm_impl

// ============================================================================
extern "C" {


int psm_solve_grid_device(psm_handle* h, const float* d_grid, int32_t n_cases, const float* out_scale,
                          float* d_fields, void* stream) {
  return solve_device(h, d_grid, n_cases, out_scale, d_fields, (hipStream_t)stream, nullptr);
}

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

int psm_ring_release(psm_handle* h, int64_t ticket) {
  if (!h) return PSM_ERR_ARG;
  psm_handle::Slot* s = nullptr;
  int rc = slot_of(h, ticket, 1, &s);                  // acquired, not submitted
  if (rc) return rc;
  s->state = 0;
  // the slot comes round again PSM_RING_SLOTS tickets later; the ticket counter does not go back (tickets stay unique)
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

}  // extern "C"
