// Device-side helpers shared by the .hip files.
#pragma once
#include <cstdint>

// One scalar load from every 64-byte line of the kernel-argument segment, all requested together at the top of a kernel.  hipcc
// fetches arguments lazily, in the basic block that first needs them: a kernel with 250-300 bytes of arguments (two argument
// structs) took three or four scalar-cache MISSES one after the other on its way to its first vector load (decode + paste: 1.9 us
// from entry to "all requests issued", 1.6 us with the lines warmed by one batch -- the later loads hit).  The convolution kernels
// (one argument struct, A/B on one box: 145.0 / 144.1 against 144.7 / 146.1 us per pass) do not use it.
// (-DPSM_NO_WARM_KERNARGS: diagnostic build without it, for A/B runs on one box.)
template <int BYTES>
__device__ __forceinline__ void psm_warm_kernargs() {
#ifdef PSM_NO_WARM_KERNARGS
  return;
#endif
  typedef const __attribute__((address_space(4))) int* kptr;
  kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  int v[(BYTES + 63) / 64];
#pragma unroll
  for (int o = 0; o < (BYTES + 63) / 64; ++o) v[o] = ka[16 * o];
#pragma unroll
  for (int o = 0; o < (BYTES + 63) / 64; ++o) asm volatile("" ::"s"(v[o]));
}

// A wave-uniform element of a read-only table that was written before the launch (block-row offsets, descriptors): through the
// scalar cache (constant address space: s_load, its own wait counter) instead of a wave-wide vector load of one value -- hipcc
// cannot prove a kernel-argument pointer read-only and takes the vector path on its own (round 6: the encode kernels' sixteen
// row-offset lookups were sixteen 64-lane loads and a vector-memory round trip in front of the rows and the basis stream).
__device__ __forceinline__ long long psm_row_base(const int64_t* table, int m) {
  typedef const __attribute__((address_space(4))) long long* ktab;
  return ((ktab)(uintptr_t)table)[__builtin_amdgcn_readfirstlane(m)];
}
