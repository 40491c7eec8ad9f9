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
// the serial offset chain, one (case, field) at a time.
//   sum/cnt : raw masked strip sums / counts for this (case, field), [n_strips]
//   tags    : blocks of the plan
//   offs    : out, [B] correction c_b subtracted from block b (NaN for skipped)
// T = float on the device, double allowed on the host.
// ---------------------------------------------------------------------------
template <typename T>
PSM_HD inline T psm_nan() { return (T)NAN; }

template <typename T, typename SR>
PSM_HD inline void psm_chain(const PsmChainParams& P, const PsmBlock* blk, const SR& sr, int field, T* up, T* offs) {
  const int n_x = P.n_x, n_y = P.n_y, NS = P.NS;
  for (int j = 0; j <= n_x && j < PSM_MAX_COLS; ++j) up[j] = (T)0;
  T c_prev = (T)0;
  // chapter5 state (PM:373-378)
  T up_m1 = psm_nan<T>(), ant0 = psm_nan<T>(), alter = (T)0;
  for (int b = 0; b < P.B; ++b) {
    const int ti = blk[b].ti, tj = blk[b].tj, sl = b * NS;
    if (blk[b].skip) { offs[b] = psm_nan<T>(); continue; }
    T c;
    if (P.variant == PSMV_DELTAS) {
      const T side_ov = sr.mean(sl + D_CUR_R_OV) - (sr.mean(sl + D_PREV_L_OV) - c_prev);
      const T side_lim = sr.mean(sl + D_CUR_R_LIM) - (sr.mean(sl + D_PREV_L_LIM) - c_prev);
      if (ti == 0) {                                            // SMD:228-246
        c = (b == 0) ? sr.mean(sl + D_COL_LAST) - (T)P.ref_bc : side_ov;
        if (tj == 0) c = side_lim;
        up[tj] = sr.mean(sl + D_BOTTOM) - c;
      } else if (ti != n_y + 1) {                               // SMD:249-283
        if (up[tj] != up[tj]) {
          if (tj == 0) c = side_lim;
          else if (tj == n_x) c = sr.mean(sl + D_TOP) - up[tj];
          else c = side_ov;
        } else {
          c = sr.mean(sl + D_TOP) - up[tj];
        }
        up[tj] = sr.mean(sl + D_BOTTOM) - c;
        if (ti == n_y) up[tj] = sr.mean(sl + D_ROWS_PI) - c;
      } else {                                                  // SMD:286-328
        if (tj == n_x) {
          c = sr.mean(sl + D_ROWS_UP) - up[tj];
        } else {
          const T n_up = sr.count(sl + D_ROWS_UP);
          if (n_up / (T)(128 * 128) > (T)0.9) c = (tj == 0) ? side_lim : side_ov;   // SMD:307
          else c = sr.mean(sl + D_ROWS_HEAD) - up[tj];
        }
      }
    } else if (P.variant == PSMV_GRADP) {
      const T side_ov = sr.mean(sl + G_CUR_L_OV) - (sr.mean(sl + G_PREV_R_OV) - c_prev);
      const T side_lim = sr.mean(sl + G_CUR_L_LIM) - (sr.mean(sl + G_PREV_R_LIM) - c_prev);
      if (ti == 0) {                                            // UGP:288-312
        if (b == 0) {
          if (field == 0) {                                     // dp_dx: first column holding a flow cell
            c = psm_nan<T>();
            for (int col = 0; col < P.S; ++col)
              if (sr.count(P.col_base + col) > (T)0) { c = sr.mean(P.col_base + col) - (T)P.ref_bc; break; }
          } else {
            c = sr.mean(sl + G_ROW1) - (T)P.ref_bc;            // dp_dy: row 1
          }
        } else {
          c = side_ov;
        }
        if (tj == n_x) c = side_lim;
        up[tj] = sr.mean(sl + G_BOTTOM) - c;
      } else if (ti != n_y + 1) {                               // UGP:314-328
        if (up[tj] != up[tj]) c = (tj == n_x) ? side_lim : side_ov;
        else c = sr.mean(sl + G_TOP) - up[tj];
        up[tj] = sr.mean(sl + G_BOTTOM) - c;
        if (ti == n_y) up[tj] = sr.mean(sl + G_ROWS_PI) - c;
      } else {                                                  // UGP:330-341
        if (up[tj] != up[tj]) c = (tj == n_x) ? side_lim : side_ov;
        else c = sr.mean(sl + G_ROWS_UP) - up[tj];
      }
    } else {                                                    // chapter5, PM:388-445
      if (ti == 0) {
        if (tj == n_x) {
          c = sr.mean(sl + C_COLS_R) - (T)0;
          up[tj] = sr.mean(sl + C_RR) - c;
        } else if (tj == -1) {
          c = sr.mean(sl + C_COLS_C) - ant0;
          up_m1 = sr.mean(sl + C_RC) - c;
        } else {
          c = sr.mean(sl + C_COLS_R) - ant0;
          up[tj] = sr.mean(sl + C_ROWS_R) - c;
        }
        ant0 = sr.mean(sl + C_COLS_0) - c;
      } else if (ti == n_y + 1) {
        if (tj == -1) c = sr.mean(sl + C_TC) - up_m1;
        else if (up[tj] != up[tj]) c = sr.mean(sl + C_COLS_R) - alter;
        else c = sr.mean(sl + C_ROWS_T) - up[tj];
      } else {
        if (tj == -1) {
          c = sr.mean(sl + C_TOPC) - up_m1;
          up_m1 = sr.mean(sl + C_RC_UNMASKED) - c;              // PM:432 (no mask)
        } else {
          if (up[tj] != up[tj]) c = sr.mean(sl + C_COLS_R) - alter;
          else c = sr.mean(sl + C_TOP) - up[tj];
          up[tj] = sr.mean(sl + C_ROWS_R) - c;
        }
      }
      alter = sr.mean(sl + C_COLS_0) - c;                       // PM:445
    }
    offs[b] = c;
    c_prev = c;
  }
}
