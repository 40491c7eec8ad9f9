// GPU-side cost of a dependent kernel boundary: a captured chain of N kernels replayed many times.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_rw(const float* in, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1.f; }
int main() {
  float *a, *b; const int n = 1 << 20;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMemset(a, 0, n * 4);
  hipStream_t s; (void)hipStreamCreate(&s);
  for (int variant = 0; variant < 4; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    const int N = 64;
    for (int i = 0; i < N; ++i) {
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, a);
      else if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, a);
      else if (variant == 2) hipLaunchKernelGGL(k_rw, dim3(32), dim3(512), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 16384);      // 64 KB ping-pong
      else hipLaunchKernelGGL(k_rw, dim3(4096), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);                           // 4 MB ping-pong
    }
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    const int R = 50;
    for (int r = 0; r < R; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("variant %d: %.2f us per kernel in a dependent chain\n", variant, us / (R * N));
  }
  return 0;
}
