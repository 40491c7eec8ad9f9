// Which XCD does workgroup i of a launch run on?  (HW_REG_XCC_ID, gfx950.)  hipcc --offload-arch=gfx950 tools/attic/xcc_probe.hip -o xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(x & 0xf);
  }
}
int main() {
  int* d; hipMalloc(&d, 4096 * sizeof(int));
  const dim3 grids[4] = {dim3(256), dim3(64), dim3(32, 2), dim3(32, 2, 3)};
  for (int rep = 0; rep < 2; ++rep)
    for (const dim3& g : grids) {
      const int n = g.x * g.y * g.z;
      hipLaunchKernelGGL(probe, g, dim3(256), 0, 0, d);
      std::vector<int> h(n);
      hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
      int ok = 0;
      for (int i = 0; i < n; ++i) ok += h[i] == i % 8;
      printf("grid (%u,%u,%u): %d of %d workgroups on XCD (linear id %% 8); first 16:", g.x, g.y, g.z, ok, n);
      for (int i = 0; i < 16 && i < n; ++i) printf(" %d", h[i]);
      printf("\n");
    }
  return 0;
}
