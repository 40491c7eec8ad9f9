// How many independent vector instructions fit behind one MFMA of a dependent chain (v_mfma_f32_32x32x16_bf16, 8 passes) on gfx950?
// Per variant: NV v_fma_f32 per MFMA (inline asm, kept in program order), 1 or 2 waves per SIMD; prints cycles per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o tools/_bin/mfma_shadow && tools/_bin/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int TWO_CHAINS>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
  f32x16 c = {0}, c2 = {0};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = (float)threadIdx.x + j;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (TWO_CHAINS && (m & 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c[j] + c2[j];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int TC>
void run(int wg_per_cu, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, TC><<<256 * wg_per_cu, 256>>>(out, cyc, 10);
  hipEventRecord(e0);
  k<NV, TC><<<256 * wg_per_cu, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("NV=%2d chains=%d waves/SIMD=%d: %7.1f ns per MFMA per wave (event), counter %lld ticks / %d MFMAs\n", NV, TC + 1, wg_per_cu, 1e6 * ms / (iters * 8), c, iters * 8);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 256 * 4); hipMalloc(&cyc, 8);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(w, out, cyc); run<2, 0>(w, out, cyc); run<4, 0>(w, out, cyc); run<6, 0>(w, out, cyc); run<7, 0>(w, out, cyc); run<8, 0>(w, out, cyc);
    run<12, 0>(w, out, cyc); run<16, 0>(w, out, cyc); run<0, 1>(w, out, cyc); run<8, 1>(w, out, cyc); run<16, 1>(w, out, cyc);
  }
  return 0;
}
