// psm_alloc.h -- device allocations of the library.  Normally hipMalloc / hipFree.  With PSM_GUARD_PAGES=1 (=2: the mirror
// image, unmapped granule in FRONT of a buffer that starts with its mapping -- accesses before the start) in the
// environment (diagnostic; read once) every allocation gets its own virtual-address reservation with an UNMAPPED granule
// behind it and the buffer is placed at the END of the mapped part (16-byte granularity), so a kernel that reads or writes
// past the end of any library buffer takes a GPU page fault at that instruction instead of silently touching a neighbour;
// freed ranges are never handed out again, so a use after free faults too.  NOTE: in guard mode (kind 1) a buffer is only
// 16-byte aligned (it ends where the mapping ends), not 256-byte like hipMalloc's: kernels must not assume more than that.
// the GPU address sanitizer is not available on this pool, this is its stand-in for the out-of-bounds-past-the-end class.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

hipError_t psm_dev_malloc(void** p, size_t bytes);
// Synchronous copies between device memory and ORDINARY (pageable) host memory, through a pinned bounce buffer owned by the
// library (one per device, with its own mutex: handles on different GPUs do not serialise each other).  hipMemcpy on pageable memory lets the runtime pin the caller's pages for the DMA engine on the fly (copies of
// 1 MiB and more) and cache those mappings by address; in a long-lived process whose heap addresses get reused that path
// produced "Memory access fault ... Write access to a read-only page" on a HOST address during a device-to-host copy
// (seen twice in ten runs of the GPU test suite, never with these).  With the bounce buffer the GPU only ever touches memory
// this library allocated with hipHostMalloc or that the caller registered explicitly.  Both return when the copy is done;
// like hipMemcpy they do not order themselves against the library's non-blocking streams.
hipError_t psm_copy_h2d(void* dst_dev, const void* src_host, size_t bytes);
hipError_t psm_copy_d2h(void* dst_host, const void* src_dev, size_t bytes);
hipError_t psm_copy_d2h_2d(void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height);
hipError_t psm_dev_free(void* p);
extern "C" int psm_debug_guard_pages(void);                       // 1 when PSM_GUARD_PAGES=1 took effect
extern "C" int psm_debug_malloc(void** p, size_t bytes);          // the same allocator for test buffers (tests/hipmem.py)
extern "C" int psm_debug_free(void* p);
extern "C" int psm_debug_copy_to_device(void* dst_dev, const void* src_host, size_t bytes);   // psm_copy_h2d / _d2h for test buffers
extern "C" int psm_debug_copy_to_host(void* dst_host, const void* src_dev, size_t bytes);
