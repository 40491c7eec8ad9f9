// psm_plan.cpp -- host-side geometry of the block pipeline (no GPU calls).
// See psm_plan.h for the reference lines each piece follows.
#include "psm_plan.h"

#include <algorithm>
#include <cstring>

#include "../../include/psm.h"

int psm_default_overlap(int variant, int S) {
  // PM:304 int(0.1*shape); SMD:788 / entry_point.py:93 overlap_ratio 0.25; UGP:708 int(0.75*shape)
  switch (variant) {
    case PSMV_CHAPTER5: return (int)(0.1 * S);
    case PSMV_DELTAS: return (int)(0.25 * S);
    case PSMV_GRADP: return (int)(0.75 * S);
  }
  return -1;
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

int psm_build_layout(int variant, int Ny, int Nx, int S, int ov, std::vector<PsmBlock>& blocks,
                     int& n_x, int& n_y, std::string& err) {
  blocks.clear();
  if (variant < 0 || variant > 2) { err = "unknown variant"; return PSM_ERR_ARG; }
  if (S <= 0 || (S % 64) != 0) { err = "block edge must be a positive multiple of 64"; return PSM_ERR_ARG; }
  if (ov <= 0) ov = psm_default_overlap(variant, S);
  if (ov <= 0 || ov >= S) { err = "overlap must lie in (0, block)"; return PSM_ERR_ARG; }
  if (Ny < S || Nx < S) { err = "grid smaller than one block"; return PSM_ERR_ARG; }
  const int st = S - ov;
  if (variant == PSMV_CHAPTER5) {                // PM:306-329
    n_x = (Nx - S) / st;
    n_y = (Ny - S) / st;
    for (int i = 0; i < n_y + 2; ++i) {
      const int y0 = (i == n_y + 1) ? Ny - S : i * st;
      for (int j = 0; j <= n_x; ++j) {
        blocks.push_back({y0, Nx - S - j * st, i, n_x - j, 0});
        if (j == n_x) blocks.push_back({y0, 0, i, -1, 0});
      }
    }
  } else if (variant == PSMV_DELTAS) {           // SMD:461-479
    n_x = ceil_div(Nx - S, st);
    n_y = (Ny - S) / st;
    for (int i = 0; i < n_y + 2; ++i) {
      const int y0 = (i == n_y + 1) ? Ny - S : i * st;
      for (int j = 0; j <= n_x; ++j)
        blocks.push_back({y0, (j == n_x) ? 0 : Nx - S - j * st, i, n_x - j, 0});
    }
  } else {                                       // UGP:479-500
    n_x = ceil_div(Nx - S, st);
    n_y = (Ny - S) / st;
    for (int i = 0; i < n_y + 2; ++i) {
      const int y0 = (i == n_y + 1) ? Ny - S : i * st;
      for (int j = 0; j <= n_x; ++j)
        blocks.push_back({y0, (j == n_x) ? Nx - S : j * st, i, j, 0});
    }
  }
  if (n_x + 1 > PSM_MAX_COLS) { err = "too many block columns"; return PSM_ERR_UNSUPPORTED; }
  return PSM_OK;
}

namespace {
struct Paste { int b, dy0, dy1, dx0, dx1, sr0, sc0; };

void add_strip(std::vector<PsmStrip>& v, size_t slot, int data, int mask, int r0, int r1, int c0, int c1, int S) {
  // python slice semantics: clamp, empty when reversed
  r0 = std::max(0, std::min(S, r0)); r1 = std::max(0, std::min(S, r1));
  c0 = std::max(0, std::min(S, c0)); c1 = std::max(0, std::min(S, c1));
  if (r1 < r0) r1 = r0;
  if (c1 < c0) c1 = c0;
  if (data < 0) { r0 = r1 = c0 = c1 = 0; data = 0; }
  v[slot] = {data, mask, r0, r1, c0, c1};
}
}  // namespace

int psm_build_plan(int variant, int Ny, int Nx, int S, int ov, bool strict, PsmPlan& plan, std::string& err) {
  if (ov <= 0) ov = psm_default_overlap(variant, S);
  int n_x = 0, n_y = 0;
  int rc = psm_build_layout(variant, Ny, Nx, S, ov, plan.blocks, n_x, n_y, err);
  if (rc != PSM_OK) return rc;
  const int st = S - ov, B = (int)plan.blocks.size();
  PsmChainParams& P = plan.cp;
  P.variant = variant; P.S = S; P.ov = ov; P.n_x = n_x; P.n_y = n_y; P.B = B;
  P.ref_bc = 0.f; P.col_base = -1;
  plan.Ny = Ny; plan.Nx = Nx;
  if (variant == PSMV_CHAPTER5) {
    P.p_i = Ny - (S * (n_y + 1) - n_y * ov);        // PM:410 "p"
    P.p_j = (Nx - S) - n_x * S + n_x * ov;          // PM:397
    P.lim = 0; P.NS = C_NS;
  } else if (variant == PSMV_DELTAS) {
    if (n_x < 1) { err = "deltas reassembly needs at least two block columns (SM_call.py:237-240)"; return PSM_ERR_UNSUPPORTED; }
    P.p_i = Ny - (st * n_y + S);                    // SMD:213
    P.p_j = Nx - (st * n_x + S);                    // SMD:216
    P.lim = ov - P.p_j;                             // SMD:238
    P.NS = D_NS;
    if (P.p_i == 0 && strict) { err = "reference undefined: p_i == 0 raises a broadcast error (SM_call.py:335)"; return PSM_ERR_UNSUPPORTED; }
  } else {
    if (n_x < 1) { err = "gradp reassembly needs at least two block columns (Eval_dual_Dense_onlycil.py:307-310)"; return PSM_ERR_UNSUPPORTED; }
    P.p_i = Ny - (S * (n_y + 1) - n_y * ov);        // UGP:277
    P.p_j = (Nx - S) - n_x * st;                    // UGP:278
    P.lim = ov - P.p_j;                             // UGP:308
    P.NS = G_NS;
  }
  const bool skip_last = (variant != PSMV_CHAPTER5) && P.p_i == 0 && !strict;
  for (auto& b : plan.blocks) b.skip = (skip_last && b.ti == n_y + 1) ? 1 : 0;
  P.skip_last = skip_last ? 1 : 0;

  // ---- strips ---------------------------------------------------------------
  const int NS = P.NS;
  plan.strips.assign((size_t)B * NS + (variant == PSMV_GRADP ? S : 0), PsmStrip{0, 0, 0, 0, 0, 0});
  for (int b = 0; b < B; ++b) {
    const size_t sl = (size_t)b * NS;
    const int prev = b - 1;  // "old_pred_field" is always the previously enumerated block
    auto& v = plan.strips;
    if (plan.blocks[b].skip) continue;
    if (variant == PSMV_DELTAS) {
      const int pi = P.p_i, lim = P.lim;
      add_strip(v, sl + D_COL_LAST, b, b, 0, S, S - 1, S, S);
      add_strip(v, sl + D_CUR_R_OV, b, b, 0, S, S - ov, S, S);
      add_strip(v, sl + D_PREV_L_OV, prev, b, 0, S, 0, ov, S);
      add_strip(v, sl + D_CUR_R_LIM, b, b, 0, S, S - lim, S, S);
      add_strip(v, sl + D_PREV_L_LIM, prev, b, 0, S, 0, lim, S);
      add_strip(v, sl + D_BOTTOM, b, b, S - ov, S, 0, S, S);
      add_strip(v, sl + D_TOP, b, b, 0, ov, 0, S, S);
      add_strip(v, sl + D_ROWS_PI, b, b, pi, S, 0, S, S);
      add_strip(v, sl + D_ROWS_UP, b, b, S - pi - ov, S - pi, 0, S, S);
      add_strip(v, sl + D_ROWS_HEAD, b, b, 0, S - pi, 0, S, S);
    } else if (variant == PSMV_GRADP) {
      const int pi = P.p_i, lim = P.lim;
      add_strip(v, sl + G_ROW1, b, b, 1, 2, 0, S, S);
      add_strip(v, sl + G_CUR_L_OV, b, b, 0, S, 0, ov, S);
      add_strip(v, sl + G_PREV_R_OV, prev, b, 0, S, S - ov, S, S);
      add_strip(v, sl + G_CUR_L_LIM, b, b, 0, S, 0, lim, S);
      add_strip(v, sl + G_PREV_R_LIM, prev, b, 0, S, S - lim, S, S);
      add_strip(v, sl + G_BOTTOM, b, b, S - ov, S, 0, S, S);
      add_strip(v, sl + G_TOP, b, b, 0, ov, 0, S, S);
      add_strip(v, sl + G_ROWS_PI, b, b, pi, S, 0, S, S);
      // python slice [-p_i-ov : -p_i]: empty when p_i == 0 (strict mode only; else that row is skipped)
      if (pi != 0) add_strip(v, sl + G_ROWS_UP, b, b, S - pi - ov, S - pi, 0, S, S);
      else add_strip(v, sl + G_ROWS_UP, b, b, 0, 0, 0, 0, S);
    } else {
      const int av = ov, p = P.p_i, pj = P.p_j;
      const int R0 = S - av, T0 = S - p - av, T1 = S - p;
      add_strip(v, sl + C_COLS_R, b, b, 0, S, R0, S, S);
      add_strip(v, sl + C_RR, b, b, R0, S, R0, S, S);
      add_strip(v, sl + C_COLS_C, b, b, 0, S, pj, pj + av, S);
      add_strip(v, sl + C_RC, b, b, R0, S, pj, pj + av, S);
      add_strip(v, sl + C_ROWS_R, b, b, R0, S, 0, S, S);
      add_strip(v, sl + C_COLS_0, b, b, 0, S, 0, av, S);
      add_strip(v, sl + C_TC, b, b, T0, T1, pj, pj + av, S);
      add_strip(v, sl + C_ROWS_T, b, b, T0, T1, 0, S, S);
      add_strip(v, sl + C_TOPC, b, b, 0, av, pj, pj + av, S);
      add_strip(v, sl + C_RC_UNMASKED, b, -1, R0, S, pj, pj + av, S);
      add_strip(v, sl + C_TOP, b, b, 0, av, 0, S, S);
    }
  }
  if (variant == PSMV_GRADP) {
    P.col_base = B * NS;
    for (int c = 0; c < S; ++c) add_strip(plan.strips, (size_t)P.col_base + c, 0, 0, 0, S, c, c + 1, S);
  }

  // ---- owner map: replay the pastes in enumeration order ---------------------
  plan.owner.assign((size_t)Ny * Nx, -1);
  for (int b = 0; b < B; ++b) {
    const PsmBlock& k = plan.blocks[b];
    if (k.skip) continue;
    Paste p{b, 0, 0, 0, 0, 0, 0};
    const bool last = (k.ti == n_y + 1);
    if (variant == PSMV_DELTAS) {                 // SMD:334-348
      const int jr = n_x - k.tj;
      if (k.tj == 0) { p.dx0 = 0; p.dx1 = S; } else { p.dx0 = Nx - S - jr * st; p.dx1 = Nx - jr * st; }
      if (last) { p.dy0 = Ny - P.p_i; p.dy1 = Ny; p.sr0 = S - P.p_i; }
      else { p.dy0 = st * k.ti; p.dy1 = p.dy0 + S; p.sr0 = 0; }
      p.sc0 = 0;
    } else if (variant == PSMV_GRADP) {           // UGP:345-356
      if (last) { p.dy0 = Ny - st; p.dy1 = Ny; p.sr0 = ov; } else { p.dy0 = k.ti * st; p.dy1 = p.dy0 + S; p.sr0 = 0; }
      if (k.tj == n_x) { p.dx0 = Nx - P.lim; p.dx1 = Nx; p.sc0 = S - P.lim; }
      else { p.dx0 = k.tj * st; p.dx1 = p.dx0 + S; p.sc0 = 0; }
    } else {                                      // PM:449-467
      if (last) { p.dy0 = Ny - st; p.dy1 = Ny; p.sr0 = ov; } else { p.dy0 = k.ti * st; p.dy1 = p.dy0 + S; p.sr0 = 0; }
      p.sc0 = 0;
      if (k.tj == -1) { p.dx0 = 0; p.dx1 = last ? Nx - (n_x + 1) * st - ov : S; }
      else { const int j = n_x - k.tj; p.dx0 = Nx - S - j * st; p.dx1 = Nx - j * st; }
    }
    for (int y = std::max(0, p.dy0); y < std::min(Ny, p.dy1); ++y)
      for (int x = std::max(0, p.dx0); x < std::min(Nx, p.dx1); ++x) {
        const int r = p.sr0 + (y - p.dy0), c = p.sc0 + (x - p.dx0);
        if (r < 0 || r >= S || c < 0 || c >= S) continue;
        plan.owner[(size_t)y * Nx + x] = (b * S + r) * S + c;
      }
  }

  // ---- global shift lists (PM:472, SMD:350, UGP:359,361) ---------------------
  for (int f = 0; f < 2; ++f) { plan.shiftA[f].clear(); plan.shiftB[f].clear(); }
  if (variant == PSMV_GRADP) {
    for (int y = 0; y < Ny; ++y) { plan.shiftA[0].push_back(y * Nx + 0); plan.shiftB[0].push_back(y * Nx + 1); }
    for (int x = 0; x < Nx; ++x) { plan.shiftA[1].push_back(1 * Nx + x); plan.shiftB[1].push_back(2 * Nx + x); }
  } else {
    for (int y = 0; y < Ny; ++y) { plan.shiftA[0].push_back(y * Nx + Nx - 1); plan.shiftB[0].push_back(y * Nx + Nx - 2); }
  }
  return PSM_OK;
}
