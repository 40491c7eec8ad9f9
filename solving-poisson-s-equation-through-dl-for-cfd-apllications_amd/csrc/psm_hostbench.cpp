// psm_hostbench.cpp -- psm_bench_host: host-buffer throughput of the surrogate measured from a C++ loop that uses the
// PUBLIC C-ABI only (include/psm.h), i.e. exactly what a C++ solver calling the library would execute per step.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/psm.h"

extern "C" int psm_bench_host(psm_handle* h, const float* grids, int32_t n_inputs, int32_t n_cases, int32_t mode, int32_t depth,
                              int32_t steps, int32_t warmup, double* seconds, float* last_fields) {
  if (!h || !grids || !seconds || n_inputs < 1 || n_cases < 1 || steps < 1 || warmup < 0 || mode < 0 || mode > 3) return PSM_ERR_ARG;
  if (depth < 1) depth = 1;
  if (depth > PSM_RING_SLOTS) depth = PSM_RING_SLOTS;
  int32_t shp[4];
  int rc = psm_grid_shape(h, shp);
  if (rc) return rc;
  const size_t gin = (size_t)n_cases * shp[0] * shp[1] * shp[2], gout = (size_t)n_cases * shp[0] * shp[1] * shp[3];
  std::vector<float> out_pageable((size_t)PSM_RING_SLOTS * gout);
  float* outs = out_pageable.data();
  if (mode == 2) {
    if ((rc = psm_host_register(h, (void*)grids, (size_t)n_inputs * gin * sizeof(float)))) return rc;
    if ((rc = psm_host_register(h, outs, out_pageable.size() * sizeof(float)))) { psm_host_unregister(h, (void*)grids); return rc; }
  }
  // mode 3 packs each slot during the warm-up: at least one full turn of the ring
  const int wu = (mode == 3 && warmup < PSM_RING_SLOTS) ? PSM_RING_SLOTS : warmup;
  std::vector<int64_t> ticket((size_t)steps + wu, -1);
  std::vector<float*> slot_out(PSM_RING_SLOTS, nullptr);
  const float* last = nullptr;
  // PSM_BENCH_VERBOSE=1: where the calling thread spends its time (submission calls / waits), to stderr
  const bool verbose = std::getenv("PSM_BENCH_VERBOSE") != nullptr;
  double t_submit = 0.0, t_wait = 0.0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  auto run = [&](int first, int count) -> int {
    int r = PSM_OK;
    if (mode == 0) {
      for (int i = first; i < first + count && r == PSM_OK; ++i)
        r = psm_solve_grid(h, grids + (size_t)(i % n_inputs) * gin, n_cases, nullptr, outs);
      last = outs;
      return r;
    }
    for (int i = first; i < first + count + depth && r == PSM_OK; ++i) {
      const int w = i - depth;                            // ticket to retire before the next submission
      const auto tw0 = now();
      if (w >= first) {
        float* o = outs + (size_t)(w % PSM_RING_SLOTS) * gout;
        if (mode == 1) { r = psm_wait_grid(h, ticket[w], o); last = o; }
        else if (mode == 2) { r = psm_wait_grid(h, ticket[w], nullptr); last = o; }
        else { r = psm_ring_wait(h, ticket[w]); last = slot_out[ticket[w] % PSM_RING_SLOTS]; }
      }
      const auto tw1 = now();
      t_wait += us(tw0, tw1);
      if (r != PSM_OK || i >= first + count) continue;
      const float* g = grids + (size_t)(i % n_inputs) * gin;
      if (mode == 1) r = psm_submit_grid(h, g, n_cases, nullptr, &ticket[i]);
      else if (mode == 2) r = psm_submit_grid_io(h, g, n_cases, nullptr, outs + (size_t)(i % PSM_RING_SLOTS) * gout, &ticket[i]);
      else {
        float *gi = nullptr, *fo = nullptr;
        r = psm_ring_acquire(h, &ticket[i], &gi, &fo);
        if (r == PSM_OK) {
          slot_out[ticket[i] % PSM_RING_SLOTS] = fo;
          if (first == 0 && i < PSM_RING_SLOTS) std::memcpy(gi, g, gin * sizeof(float));   // pack once per slot, untimed
          r = psm_ring_submit(h, ticket[i], n_cases, nullptr);
        }
      }
      t_submit += us(tw1, now());
    }
    return r;
  };
  if (wu > 0) rc = run(0, wu);
  if (rc == PSM_OK) {
    const auto t0 = std::chrono::steady_clock::now();
    t_submit = t_wait = 0.0;
    rc = run(wu, steps);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (verbose && mode != 0)
      std::fprintf(stderr, "psm_bench_host mode %d depth %d: %.1f us per solve = %.1f us in submission calls + %.1f us waiting\n",
                   mode, depth, *seconds * 1e6 / steps, t_submit / steps, t_wait / steps);
  }
  if (rc != PSM_OK && mode != 0) {
    // a failed run must not leave slots taken: wait for every ticket that is still in flight (newest PSM_RING_SLOTS at most)
    // and give back one that was acquired but not submitted; the first error stays the result
    const int issued = (int)ticket.size();
    for (int i = issued > PSM_RING_SLOTS ? issued - PSM_RING_SLOTS : 0; i < issued; ++i) {
      if (ticket[i] < 0) continue;
      if (mode == 3) { if (psm_ring_wait(h, ticket[i]) != PSM_OK) (void)psm_ring_release(h, ticket[i]); }
      else (void)psm_wait_grid(h, ticket[i], mode == 1 ? outs + (size_t)(i % PSM_RING_SLOTS) * gout : nullptr);
    }
  }
  if (rc == PSM_OK && last_fields && last) std::memcpy(last_fields, last, gout * sizeof(float));
  if (mode == 2) { psm_host_unregister(h, (void*)grids); psm_host_unregister(h, outs); }
  return rc;
}
