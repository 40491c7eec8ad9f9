// psm_launch.h -- PSM_LAUNCH: kernel launches that can be stamped dispatch by dispatch (psm_time_kernels, psm_unet_time_kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <vector>

// Dispatch-level timing of every kernel launch of the solve path (psm_time_kernels): while a probe is installed on the
// calling thread, PSM_LAUNCH stamps each dispatch's own begin / end into an event pair (hipExtLaunchKernelGGL -- the
// source rocprofv3 reads, not marker packets around the launch) and records the kernel's name.
struct PsmLaunchProbe {
  struct Rec { const char* name; hipEvent_t e0, e1; int tag; };
  int tag = -1;                          // set by the caller around a launch (psm_unet_time_kernels: the convolution index)
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;          // recycled events
  hipEvent_t get() { if (pool.empty()) { hipEvent_t e; (void)hipEventCreate(&e); return e; } hipEvent_t e = pool.back(); pool.pop_back(); return e; }
};
extern thread_local PsmLaunchProbe* psm_launch_probe;
#define PSM_LAUNCH(kern, grid, block, lds, st, ...)                                                                   \
  do {                                                                                                                \
    if (psm_launch_probe) {                                                                                           \
      PsmLaunchProbe::Rec r_{#kern, psm_launch_probe->get(), psm_launch_probe->get(), psm_launch_probe->tag};                                \
      psm_launch_probe->recs.push_back(r_);                                                                           \
      hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)(lds), st, r_.e0, r_.e1, 0, __VA_ARGS__);               \
    } else {                                                                                                          \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                    \
    }                                                                                                                 \
  } while (0)

