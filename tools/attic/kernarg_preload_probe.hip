// What does a dependent launch cost with the kernel's arguments preloaded into SGPRs by the command processor (gfx950:
// -mllvm -amdgpu-kernarg-preload-count=N, flat arguments only) against the ordinary s_load of the kernarg segment?
// A chain of small dependent kernels (each reads 64 KB the previous one wrote and writes 64 KB), plain launches on one stream.
//   hipcc -O3 --offload-arch=gfx950 tools/attic/kernarg_preload_probe.hip -o tools/_bin/kp_plain
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/attic/kernarg_preload_probe.hip -o tools/_bin/kp_preload
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Args { const float* in; float* out; int n; float s; };
__global__ __launch_bounds__(256) void k_flat(const float* in, float* out, int n, float s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}
__global__ __launch_bounds__(256) void k_struct(Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) a.out[i] = a.in[i] * a.s;
}
__global__ void k_empty() {}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int n = 16384, chain = 4000;
  float *a, *b;
  CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&b, n * 4));
  CHK(hipMemset(a, 0, n * 4)); CHK(hipMemset(b, 0, n * 4));
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipStreamSynchronize(st));
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < chain; ++i) {
        float* x = (i & 1) ? b : a; float* y = (i & 1) ? a : b;
        if (mode == 0) hipLaunchKernelGGL(k_flat, dim3(n / 256), dim3(256), 0, st, x, y, n, 1.0f);
        else if (mode == 1) { Args g{x, y, n, 1.0f}; hipLaunchKernelGGL(k_struct, dim3(n / 256), dim3(256), 0, st, g); }
        else hipLaunchKernelGGL(k_empty, dim3(n / 256), dim3(256), 0, st);
      }
      CHK(hipStreamSynchronize(st));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
      if (us < best) best = us;
    }
    printf("%s: %.3f us per dependent launch\n", mode == 0 ? "flat arguments  " : mode == 1 ? "struct by value " : "empty kernel    ", best);
  }
  return 0;
}
