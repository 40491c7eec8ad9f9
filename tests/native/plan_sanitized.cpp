// CPU-only check of the host planner (csrc/psm_plan.cpp) under AddressSanitizer + UBSan: builds the plan
// of many grid shapes for every variant, replays the serial offset chain on synthetic strip means and
// checks the structural invariants the device code relies on (indices inside their tables).
// Built and run by tests/test_plan_sanitized.py with g++ -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "psm_plan.h"

int main() {
  const int S = 128;
  int n_ok = 0, n_refused = 0;
  std::mt19937 rng(7);
  for (int variant = 0; variant < 3; ++variant) {
    for (int trial = 0; trial < 50; ++trial) {
      const int Ny = 100 + (int)(rng() % 700), Nx = 100 + (int)(rng() % 3200);
      for (int strict = 0; strict < 2; ++strict) {
        PsmPlan plan;
        std::string err;
        const int rc = psm_build_plan(variant, Ny, Nx, S, 0, strict != 0, plan, err);
        if (rc != 0) {
          if (err.empty()) { std::printf("refused without a message: v=%d %dx%d\n", variant, Ny, Nx); return 1; }
          ++n_refused;
          continue;
        }
        const int B = (int)plan.blocks.size(), NS = plan.cp.NS;
        if (B != plan.cp.B || B < 1) { std::printf("block count mismatch\n"); return 1; }
        for (const PsmBlock& b : plan.blocks)
          if (b.y0 < 0 || b.x0 < 0 || b.y0 + S > Ny || b.x0 + S > Nx) { std::printf("block outside the grid\n"); return 1; }
        if ((int)plan.strips.size() < B * NS) { std::printf("strip table too small\n"); return 1; }
        for (const PsmStrip& s : plan.strips) {
          if (s.data < 0 || s.data >= B || s.mask >= B) { std::printf("strip block index out of range\n"); return 1; }
          if (s.r0 < 0 || s.c0 < 0 || s.r1 > S || s.c1 > S) { std::printf("strip rectangle outside the block\n"); return 1; }
        }
        if ((int)plan.owner.size() != Ny * Nx) { std::printf("owner map size\n"); return 1; }
        for (int32_t o : plan.owner)
          if (o < -1 || o >= B * S * S) { std::printf("owner out of range\n"); return 1; }
        for (int f = 0; f < 2; ++f) {
          if (plan.shiftA[f].size() != plan.shiftB[f].size()) { std::printf("shift lists differ\n"); return 1; }
          for (size_t k = 0; k < plan.shiftA[f].size(); ++k)
            if (plan.shiftA[f][k] < 0 || plan.shiftA[f][k] >= Ny * Nx || plan.shiftB[f][k] < 0 || plan.shiftB[f][k] >= Ny * Nx) {
              std::printf("shift index outside the grid\n"); return 1;
            }
        }
        // serial chain on synthetic means (some NaN: empty strips), through the array context the device fallback uses
        if (plan.cp.n_x + 2 <= PSM_MAX_COLS) {
          const int nst = (int)plan.strips.size();
          std::vector<float> mean(nst), cnt(nst), up(PSM_MAX_COLS, 0.f), offs(B, 0.f);
          for (int e = 0; e < nst; ++e) {
            const bool empty = (rng() % 17) == 0;
            mean[e] = empty ? NAN : (float)((int)(rng() % 2001) - 1000) * 1e-3f;
            cnt[e] = empty ? 0.f : 100.f;
          }
          PsmArrayChainCtx<float> cx{plan.blocks.data(), mean.data(), cnt.data(), NS, plan.cp.col_base, S, up.data(), offs.data()};
          psm_chain<float>(plan.cp, cx, 0);
        }
        ++n_ok;
      }
    }
  }
  std::printf("plans built: %d, refused with a message: %d\n", n_ok, n_refused);
  return 0;
}
