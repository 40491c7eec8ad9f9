// psm_alloc.cpp -- see psm_alloc.h
#include "psm_alloc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct Guarded { void* va; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; };
std::mutex g_mu;
std::unordered_map<void*, Guarded> g_live;

int guard_kind() {                                                 // 0 off, 1 unmapped granule BEHIND the buffer, 2 in FRONT of it
  static const int k = [] { const char* e = getenv("PSM_GUARD_PAGES"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
  return k;
}
bool guard_mode() { return guard_kind() != 0; }
}  // namespace

hipError_t psm_dev_malloc(void** p, size_t bytes) {
  if (!guard_mode()) return hipMalloc(p, bytes);
  *p = nullptr;
  if (bytes == 0) bytes = 1;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  if (gran == 0) return hipErrorInvalidValue;
  Guarded g{};
  g.mapped = (bytes + gran - 1) / gran * gran;
  g.reserved = g.mapped + gran;                                   // one granule of reserved, never mapped, address space behind it
  void* base = nullptr;
  if ((e = hipMemAddressReserve(&base, g.reserved, gran, nullptr, 0)) != hipSuccess) return e;
  g.va = guard_kind() == 2 ? (char*)base + gran : base;          // kind 2: the unmapped granule comes first
  if ((e = hipMemCreate(&g.handle, g.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(base, g.reserved); return e; }
  if ((e = hipMemMap(g.va, g.mapped, 0, g.handle, 0)) != hipSuccess) {
    (void)hipMemRelease(g.handle); (void)hipMemAddressFree(base, g.reserved); return e;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(g.va, g.mapped, &acc, 1)) != hipSuccess) {
    (void)hipMemUnmap(g.va, g.mapped); (void)hipMemRelease(g.handle); (void)hipMemAddressFree(base, g.reserved); return e;
  }
  // PSM_GUARD_FILL=<byte>: fill the whole mapping (memory from hipMemCreate is not cleared) -- 0 mimics the fresh pages hipMalloc
  // usually hands out, 255 poisons: floats read as NaN, indices as -1, so a kernel that depends on what it never wrote shows
  if (const char* f = getenv("PSM_GUARD_FILL")) {
    if ((e = hipMemset(g.va, atoi(f) & 255, g.mapped)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
      (void)hipMemUnmap(g.va, g.mapped); (void)hipMemRelease(g.handle); (void)hipMemAddressFree(base, g.reserved); return e;
    }
  }
  const size_t used = (bytes + 15) / 16 * 16;
  *p = guard_kind() == 2 ? g.va : (char*)g.va + (g.mapped - used);   // kind 1: the buffer ends where the mapping ends; kind 2: it starts where it starts
  if (getenv("PSM_GUARD_LOG")) fprintf(stderr, "psm_alloc: + %p %zu B (va %p, mapped %zu)\n", *p, bytes, g.va, g.mapped);
  std::lock_guard<std::mutex> lk(g_mu);
  g_live[*p] = g;
  return hipSuccess;
}

hipError_t psm_dev_free(void* p) {
  if (!p) return hipSuccess;
  if (!guard_mode()) return hipFree(p);
  Guarded g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {                                     // not a live buffer of this allocator: a double free or a foreign pointer
      fprintf(stderr, "psm_alloc: free of %p, which is not a live guarded buffer (double free?)\n", p);
      return hipErrorInvalidValue;
    }
    g = it->second;
    g_live.erase(it);
  }
  if (getenv("PSM_GUARD_LOG")) fprintf(stderr, "psm_alloc: - %p (va %p, mapped %zu)\n", p, g.va, g.mapped);
  hipError_t e = hipDeviceSynchronize();                          // hipFree's implicit synchronisation
  (void)hipMemUnmap(g.va, g.mapped);
  (void)hipMemRelease(g.handle);
  // The address range stays reserved for the life of the process: (i) a use after free then faults as well, (ii) on ROCm 7.2 a
  // range handed out again right after hipMemAddressFree was seen to alias the physical pages of a LATER allocation (the
  // bound-pattern table read back another buffer's contents) -- a diagnostic run is short: the GPU test suite and a soak of
  // 250 autotuned networks fit, a longer one ends with hipErrorOutOfMemory from the reservations.
  return e;
}

// ---- bounce-buffer copies ---------------------------------------------------------------------------------------------
// One pinned bounce buffer and one mutex PER DEVICE (handles on different GPUs upload and read back concurrently; threads on
// the same GPU take turns).  The copies themselves stay legacy-stream hipMemcpy on purpose: like hipMemcpy they are ordered
// after the caller's null-stream work -- psm_bind_geometry(on_device) reads a grid the caller may just have uploaded on
// that stream -- and they return when the data has arrived.
namespace {
constexpr size_t BOUNCE_BYTES = (size_t)4 << 20;
constexpr int MAX_DEV = 64;
struct Bounce { std::mutex mu; void* buf = nullptr; };
Bounce g_bounce[MAX_DEV];
hipError_t bounce_for_current_device(Bounce** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
  *out = &g_bounce[dev];
  return hipSuccess;
}
hipError_t bounce_ready(Bounce& b) {                                           // called with b.mu held
  if (b.buf) return hipSuccess;
  return hipHostMalloc(&b.buf, BOUNCE_BYTES, hipHostMallocPortable);           // one per device, lives until exit
}
}  // namespace

hipError_t psm_copy_h2d(void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  Bounce* b = nullptr;
  hipError_t e = bounce_for_current_device(&b);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(b->mu);
  e = bounce_ready(*b);
  for (size_t off = 0; e == hipSuccess && off < bytes; off += BOUNCE_BYTES) {
    const size_t n = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
    std::memcpy(b->buf, (const char*)src_host + off, n);
    e = hipMemcpy((char*)dst_dev + off, b->buf, n, hipMemcpyHostToDevice);
  }
  return e;
}

hipError_t psm_copy_d2h(void* dst_host, const void* src_dev, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  Bounce* b = nullptr;
  hipError_t e = bounce_for_current_device(&b);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(b->mu);
  e = bounce_ready(*b);
  for (size_t off = 0; e == hipSuccess && off < bytes; off += BOUNCE_BYTES) {
    const size_t n = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
    e = hipMemcpy(b->buf, (const char*)src_dev + off, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) std::memcpy((char*)dst_host + off, b->buf, n);
  }
  return e;
}

hipError_t psm_copy_d2h_2d(void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height) {
  if (width == 0 || height == 0) return hipSuccess;
  if (width > BOUNCE_BYTES) return hipErrorInvalidValue;
  Bounce* b = nullptr;
  hipError_t e = bounce_for_current_device(&b);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(b->mu);
  e = bounce_ready(*b);
  const size_t rows_per = BOUNCE_BYTES / width;
  for (size_t r0 = 0; e == hipSuccess && r0 < height; r0 += rows_per) {
    const size_t nr = height - r0 < rows_per ? height - r0 : rows_per;
    e = hipMemcpy2D(b->buf, width, (const char*)src_dev + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost);
    for (size_t r = 0; e == hipSuccess && r < nr; ++r) std::memcpy((char*)dst_host + (r0 + r) * dpitch, (const char*)b->buf + r * width, width);
  }
  return e;
}

extern "C" int psm_debug_copy_to_device(void* dst_dev, const void* src_host, size_t bytes) { return (int)psm_copy_h2d(dst_dev, src_host, bytes); }
extern "C" int psm_debug_copy_to_host(void* dst_host, const void* src_dev, size_t bytes) { return (int)psm_copy_d2h(dst_host, src_dev, bytes); }
extern "C" int psm_debug_guard_pages(void) { return guard_mode() ? 1 : 0; }
extern "C" int psm_debug_malloc(void** p, size_t bytes) { return p ? (int)psm_dev_malloc(p, bytes) : (int)hipErrorInvalidValue; }
extern "C" int psm_debug_free(void* p) { return (int)psm_dev_free(p); }
