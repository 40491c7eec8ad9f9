// psm_plan.h -- block layout, overlap-strip table, owner map and the serial
// offset chain of the block reassembly (host side; the chain is shared with
// the device through PSM_HD).
//
// Reference semantics (paths relative to the reference repository):
//   PM  Thesis_Work/Chapter5/parallelized/test_case/python_module.py:303-332,373-472
//   SMD Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/SM_call.py:182-365,452-482
//   UGP Improved_SM/U_to_gradP/evaluation/Eval_dual_Dense_onlycil.py:255-369,476-500
//
// Key identity used for the GPU formulation: every correction subtracts one
// scalar c_b from a whole block, and mean(strip - c) = mean(strip) - c.  All
// masked strip means are therefore taken on the RAW decoded blocks in
// parallel, and the data-dependent recurrence (including the np.isnan tests
// of the reference) runs afterwards on scalars only.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PSM_HD __host__ __device__
#else
#define PSM_HD
#endif

#define PSM_MAX_COLS 512  // capacity of the BC_ups row (n_x + 1)

enum { PSMV_CHAPTER5 = 0, PSMV_DELTAS = 1, PSMV_GRADP = 2 };

// strip slots per block ------------------------------------------------------
// deltas (SMD)
enum { D_COL_LAST = 0, D_CUR_R_OV, D_PREV_L_OV, D_CUR_R_LIM, D_PREV_L_LIM, D_BOTTOM, D_TOP,
       D_ROWS_PI, D_ROWS_UP, D_ROWS_HEAD, D_NS };
// gradp (UGP)
enum { G_ROW1 = 0, G_CUR_L_OV, G_PREV_R_OV, G_CUR_L_LIM, G_PREV_R_LIM, G_BOTTOM, G_TOP,
       G_ROWS_PI, G_ROWS_UP, G_NS };
// chapter5 (PM)
enum { C_COLS_R = 0, C_RR, C_COLS_C, C_RC, C_ROWS_R, C_COLS_0, C_TC, C_ROWS_T, C_TOPC, C_RC_UNMASKED,
       C_TOP, C_NS };

struct PsmBlock {
  int32_t y0, x0, ti, tj;
  int32_t skip;  // duplicate last row left out of the reassembly (p_i == 0, non-strict)
};

struct PsmStrip {          // rectangle [r0,r1) x [c0,c1) of block `data`, masked by block `mask`'s flow cells
  int32_t data, mask;      // block indices; mask < 0: unmasked
  int32_t r0, r1, c0, c1;  // empty rectangle -> (sum 0, count 0) -> NaN mean, like np.mean([])
};

struct PsmChainParams {    // scalars the recurrence needs (device-copyable)
  int32_t variant, S, ov, n_x, n_y, B, NS;
  int32_t p_i, p_j, lim;
  int32_t col_base;        // gradp: slot of the first of S single-column strips of block 0; else -1
  int32_t skip_last;       // 1: the duplicate last block row (p_i == 0) is left out of the reassembly
  float ref_bc;
};

struct PsmPlan {
  PsmChainParams cp;
  int32_t Ny = 0, Nx = 0;
  std::vector<PsmBlock> blocks;
  std::vector<PsmStrip> strips;       // B*NS (+S for gradp)
  std::vector<int32_t> owner;         // [Ny*Nx] b*S*S + r*S + c of the last paste covering the cell
  // global shift: mean_i(3*field[A_i] - field[B_i]) / 3, lists of cell indices per output field
  std::vector<int32_t> shiftA[2], shiftB[2];
};

int psm_default_overlap(int variant, int S);
// returns 0 or a negative PSM_ERR_* code, message in err
int psm_build_layout(int variant, int Ny, int Nx, int S, int ov, std::vector<PsmBlock>& blocks,
                     int& n_x, int& n_y, std::string& err);
int psm_build_plan(int variant, int Ny, int Nx, int S, int ov, bool strict, PsmPlan& plan, std::string& err);

// ---------------------------------------------------------------------------
// the serial offset chain, one (case, field) at a time, written against a small
// context interface so that the same statements run on the host (arrays), on the
// device from LDS, and on the device from wave registers (lane-distributed state):
//   cx.tag(b, ti, tj, skip)      block tags
//   cx.mean(b, K) / cx.count(b, K)   raw masked mean / count of strip slot K of block b
//                                (mean of an empty strip is NaN, like np.mean([]))
//   cx.first_col_mean()          gradp: mean of the first column of block 0 holding a flow cell (NaN if none)
//   cx.up(j) / cx.set_up(j, v)   BC_ups row
//   cx.set_off(b, c)             correction subtracted from block b (NaN for skipped blocks)
// T = float on the device, double allowed on the host.
// ---------------------------------------------------------------------------
template <typename T>
PSM_HD inline T psm_nan() { return (T)NAN; }

// Written with selects instead of branches: on the device the recurrence is executed by a
// whole wave on uniform values, where every taken branch costs an instruction-fetch bubble;
// the NaN tests of the reference (np.isnan(BC_ups[..])) become selects on `u != u`.
template <typename T, int VARIANT, typename CX>
PSM_HD inline void psm_chain_v(const PsmChainParams& P, CX& cx, int field) {
  const int n_x = P.n_x, n_y = P.n_y;
  const T ref = (T)P.ref_bc;
  T c_prev = (T)0;
  T up_m1 = psm_nan<T>(), ant0 = psm_nan<T>(), alter = (T)0;     // chapter5 state (PM:373-378)
  for (int b = 0; b < P.B; ++b) {
    int ti, tj, skip;
    cx.tag(b, ti, tj, skip);
    if (skip) { cx.set_off(b, psm_nan<T>()); continue; }
    const bool first = (ti == 0), last = (ti == n_y + 1);
    const int jc = tj < 0 ? 0 : tj;
    const T u = cx.up(jc);
    const bool unan = (u != u);
    T c, newup;
    if (VARIANT == PSMV_DELTAS) {
      const T side_ov = cx.mean(b, D_CUR_R_OV) - (cx.mean(b, D_PREV_L_OV) - c_prev);
      const T side_lim = cx.mean(b, D_CUR_R_LIM) - (cx.mean(b, D_PREV_L_LIM) - c_prev);
      const T side = (tj == 0) ? side_lim : side_ov;
      const T top = cx.mean(b, D_TOP) - u;
      // SMD:228-246
      const T c_first = (tj == 0) ? side_lim : ((b == 0) ? cx.mean(b, D_COL_LAST) - ref : side_ov);
      // SMD:249-283
      const T c_mid = unan ? ((tj != 0 && tj == n_x) ? top : side) : top;
      // SMD:286-328 (the 0.9 test of SMD:307 is hard-wired to 128^2 cells)
      const bool use_side = cx.count(b, D_ROWS_UP) / (T)(128 * 128) > (T)0.9;
      const T c_last = (tj == n_x) ? cx.mean(b, D_ROWS_UP) - u : (use_side ? side : cx.mean(b, D_ROWS_HEAD) - u);
      c = first ? c_first : (last ? c_last : c_mid);
      const T bottom = (!first && ti == n_y) ? cx.mean(b, D_ROWS_PI) : cx.mean(b, D_BOTTOM);
      newup = last ? u : bottom - c;
    } else if (VARIANT == PSMV_GRADP) {
      const T side_ov = cx.mean(b, G_CUR_L_OV) - (cx.mean(b, G_PREV_R_OV) - c_prev);
      const T side_lim = cx.mean(b, G_CUR_L_LIM) - (cx.mean(b, G_PREV_R_LIM) - c_prev);
      const T side = (tj == n_x) ? side_lim : side_ov;
      T c_first = side;                                          // UGP:288-312
      if (b == 0) {
        c_first = (field == 0 ? cx.first_col_mean() : cx.mean(b, G_ROW1)) - ref;
        if (tj == n_x) c_first = side_lim;
      }
      const T c_mid = unan ? side : cx.mean(b, G_TOP) - u;       // UGP:314-328
      const T c_last = unan ? side : cx.mean(b, G_ROWS_UP) - u;  // UGP:330-341
      c = first ? c_first : (last ? c_last : c_mid);
      const T bottom = (!first && ti == n_y) ? cx.mean(b, G_ROWS_PI) : cx.mean(b, G_BOTTOM);
      newup = last ? u : bottom - c;
    } else {                                                     // chapter5, PM:388-445
      const bool m1 = (tj == -1), nx = (tj == n_x);
      const T colsR = cx.mean(b, C_COLS_R);
      const T c_first = nx ? colsR - (T)0 : ((m1 ? cx.mean(b, C_COLS_C) : colsR) - ant0);
      const T c_last = m1 ? cx.mean(b, C_TC) - up_m1 : (unan ? colsR - alter : cx.mean(b, C_ROWS_T) - u);
      const T c_mid = m1 ? cx.mean(b, C_TOPC) - up_m1 : (unan ? colsR - alter : cx.mean(b, C_TOP) - u);
      c = first ? c_first : (last ? c_last : c_mid);
      // BC_ups / BC_up_ / BC_ant_0 / BC_alter updates
      const T up_first = (nx ? cx.mean(b, C_RR) : cx.mean(b, C_ROWS_R)) - c;
      newup = (last || m1) ? u : (first ? up_first : cx.mean(b, C_ROWS_R) - c);
      const T m1_new = first ? cx.mean(b, C_RC) - c : cx.mean(b, C_RC_UNMASKED) - c;   // PM:400 / PM:432 (no mask)
      up_m1 = (m1 && !last) ? m1_new : up_m1;
      const T left = cx.mean(b, C_COLS_0) - c;
      ant0 = first ? left : ant0;
      alter = left;                                              // PM:445
    }
    cx.set_up(jc, newup);
    cx.set_off(b, c);
    c_prev = c;
  }
}

template <typename T, typename CX>
PSM_HD inline void psm_chain(const PsmChainParams& P, CX& cx, int field) {
  if (P.variant == PSMV_DELTAS) psm_chain_v<T, PSMV_DELTAS>(P, cx, field);
  else if (P.variant == PSMV_GRADP) psm_chain_v<T, PSMV_GRADP>(P, cx, field);
  else psm_chain_v<T, PSMV_CHAPTER5>(P, cx, field);
}

// array-backed context (host replay, device fallback from LDS)
template <typename T>
struct PsmArrayChainCtx {
  const PsmBlock* blk; const T* mean_; const T* cnt_; int NS; int col_base; int S;
  T* up_; T* offs_;
  PSM_HD void tag(int b, int& ti, int& tj, int& skip) const { ti = blk[b].ti; tj = blk[b].tj; skip = blk[b].skip; }
  PSM_HD T mean(int b, int K) const { return mean_[b * NS + K]; }
  PSM_HD T count(int b, int K) const { return cnt_[b * NS + K]; }
  PSM_HD T first_col_mean() const {
    for (int c = 0; c < S; ++c) if (cnt_[col_base + c] > (T)0) return mean_[col_base + c];
    return psm_nan<T>();
  }
  PSM_HD T up(int j) const { return up_[j]; }
  PSM_HD void set_up(int j, T v) { up_[j] = v; }
  PSM_HD void set_off(int b, T c) { offs_[b] = c; }
};
