#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_touch(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
__global__ void k_dep(const float* in, float* out, int n) {   // 2 dependent loads
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = (int)in[i] ;
  out[i] = in[(j + i) % n] + 1.f;
}
int main() {
  float *a, *b; int n = 65536;
  hipMalloc(&a, n * 4 * 64); hipMalloc(&b, n * 4 * 64);
  hipMemset(a, 0, n * 4 * 64); hipMemset(b, 0, n * 4 * 64);
  hipStream_t s; hipStreamCreate(&s);
  for (int it = 0; it < 200; ++it) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_empty, dim3(60), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_dep, dim3(256), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_dep, dim3(256), dim3(256), 0, s, b, a, n);
  }
  hipStreamSynchronize(s);
  printf("done\n");
  return 0;
}
