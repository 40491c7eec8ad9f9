// psm_alloc.h -- device allocations of the library.  Normally hipMalloc / hipFree.  With PSM_GUARD_PAGES=1 (=2: the mirror
// image, unmapped granule in FRONT of a buffer that starts with its mapping -- accesses before the start) in the
// environment (diagnostic; read once) every allocation gets its own virtual-address reservation with an UNMAPPED granule
// behind it and the buffer is placed at the END of the mapped part (16-byte granularity), so a kernel that reads or writes
// past the end of any library buffer takes a GPU page fault at that instruction instead of silently touching a neighbour;
// freed ranges are never handed out again, so a use after free faults too:
// the GPU address sanitizer is not available on this pool, this is its stand-in for the out-of-bounds-past-the-end class.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

hipError_t psm_dev_malloc(void** p, size_t bytes);
hipError_t psm_dev_free(void* p);
extern "C" int psm_debug_guard_pages(void);                       // 1 when PSM_GUARD_PAGES=1 took effect
extern "C" int psm_debug_malloc(void** p, size_t bytes);          // the same allocator for test buffers (tests/hipmem.py)
extern "C" int psm_debug_free(void* p);
