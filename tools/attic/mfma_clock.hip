// Effective shader clock under f32 MFMA load: N back-to-back v_mfma_f32_16x16x4_f32 (32 cycles of issue each on one
// SIMD, 4 independent accumulators) timed with the 100 MHz s_memrealtime, on 1 workgroup and on the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int n) {
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = 1.0001f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
  float* o; unsigned long long* t; const int NB = 2048;
  (void)hipMalloc(&o, NB * 256 * 4); (void)hipMalloc(&t, NB * 8);
  unsigned long long h[2048];
  for (int blocks : {1, 256, 512, 1024}) {
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, t, n); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, t, blocks * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int b = 0; b < blocks; ++b) mx = h[b] > mx ? h[b] : mx;
    const double us = mx * 0.01, cyc = 4.0 * n * 32;
    printf("blocks %4d: %8.1f us for %d MFMAs per wave -> %.2f GHz effective (32 cyc/MFMA), %.1f TFLOP/s chip-wide at this rate\n",
           blocks, us, 4 * n, cyc / us * 1e-3, (blocks >= 256 ? 1024.0 : 4.0 * blocks) * 4 * n * 2048 / us * 1e-6);
  }
  return 0;
}
