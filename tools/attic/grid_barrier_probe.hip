// What does a grid-wide barrier cost on MI355X next to a kernel boundary?  One launch of G resident workgroups runs N phases;
// in each phase a workgroup writes 4 KB, crosses the barrier (agent-scope release / acquire on one counter in device
// memory), then reads the 4 KB a workgroup of ANOTHER XCD wrote in that phase and checks it.  Compared with N dependent
// launches of the same body.  Every spin is bounded (a workgroup that never sees the counter sets an error flag and goes
// on), so the grid always drains.
//   hipcc -O3 --offload-arch=gfx950 tools/attic/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int WORDS = 1024;          // 4 KB per workgroup and phase
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ void body_write(unsigned* buf, int g, int phase) {
  unsigned* p = buf + (size_t)g * WORDS;
  for (int i = threadIdx.x; i < WORDS; i += blockDim.x) p[i] = (unsigned)(phase * 131071 + g * 977 + i);
}
__device__ __forceinline__ unsigned body_check(const unsigned* buf, int g, int G, int phase) {
  const int o = (g + 3) % G;          // linear id + 3: another XCD
  const unsigned* p = buf + (size_t)o * WORDS;
  unsigned bad = 0;
  for (int i = threadIdx.x; i < WORDS; i += blockDim.x) bad += p[i] != (unsigned)(phase * 131071 + o * 977 + i);
  return bad;
}

__global__ void __launch_bounds__(256) persistent(unsigned* buf0, unsigned* buf1, unsigned* counter, unsigned* err, int phases, int fence_only) {
  const int g = blockIdx.x, G = gridDim.x;
  unsigned bad = 0;
  for (int ph = 0; ph < phases; ++ph) {
    unsigned* buf = (ph & 1) ? buf1 : buf0;
    body_write(buf, g, ph);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (fence_only == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(ph + 1) * (unsigned)G;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want)
          if (++spins > SPIN_MAX) { atomicOr(err, 1u); break; }
      } else {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
      }
    }
    __syncthreads();
    if (fence_only == 0) bad += body_check(buf, g, G, ph);
  }
  if (bad) atomicOr(err, 2u);
}

__global__ void __launch_bounds__(256) one_phase(unsigned* buf, const unsigned* prev, unsigned* err, int ph) {
  const int g = blockIdx.x, G = gridDim.x;
  unsigned bad = ph > 0 ? body_check(prev, g, G, ph - 1) : 0;
  body_write(buf, g, ph);
  if (bad) atomicOr(err, 2u);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 200;
  unsigned *b0, *b1, *cnt, *err;
  CK(hipMalloc(&b0, 1024 * WORDS * 4)); CK(hipMalloc(&b1, 1024 * WORDS * 4)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent, 256, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs, %d resident workgroups of 256 per CU\n", prop.name, prop.multiProcessorCount, occ);
  for (int G : {64, 128, 256, 512}) {
    if (G > occ * prop.multiProcessorCount) continue;
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f; unsigned herr = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (mode < 2) {
          void* args[] = {&b0, &b1, &cnt, &err, (void*)&phases, &mode};
          CK(hipLaunchCooperativeKernel((const void*)persistent, dim3(G), dim3(256), args, 0, 0));
        } else {
          for (int ph = 0; ph < phases; ++ph)
            hipLaunchKernelGGL(one_phase, dim3(G), dim3(256), 0, 0, (ph & 1) ? b1 : b0, (ph & 1) ? b0 : b1, err, ph);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); herr |= h;
      }
      printf("G=%3d %-34s %7.3f us per phase  (err %u)\n", G,
             mode == 0 ? "persistent, counter barrier" : mode == 1 ? "persistent, fences only (no wait)" : "one launch per phase", 1e3f * best / phases, herr);
    }
  }
  return 0;
}
