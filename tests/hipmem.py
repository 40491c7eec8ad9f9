"""Minimal HIP runtime access through ctypes for the GPU tests (device buffers without importing
torch, whose first import on a fresh box can take minutes).  Host <-> device copies go through the library's pinned bounce
buffer (psm_debug_copy_to_device / _to_host): hipMemcpy on ordinary NumPy memory lets the HIP runtime pin the caller's pages on
the fly, a path that produced "Write access to a read-only page" faults on host addresses in this long-lived test process.
With PSM_GUARD_PAGES=1 / 2 in the environment the buffers come from the library's guard-page allocator
(csrc/psm_alloc.cpp): a kernel that runs past the end (before the start) of a test's buffer faults instead of touching a
neighbour."""
import ctypes as C
import os

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


def _psm():
    from psm_amd import _lib
    return _lib.load()


class DeviceArray:
    def __init__(self, host: np.ndarray = None, shape=None, dtype=np.float32):
        if host is not None:
            host = np.ascontiguousarray(host)
            shape, dtype = host.shape, host.dtype
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        self._guarded = os.environ.get("PSM_GUARD_PAGES") in ("1", "2")
        if self._guarded:
            lib = _psm()
            assert lib.psm_debug_guard_pages() == 1 and lib.psm_debug_malloc(C.byref(p), self.nbytes) == 0
            self._lib = lib
        else:
            assert hip().hipMalloc(C.byref(p), self.nbytes) == 0
        self.ptr = p.value
        if host is not None:
            assert _psm().psm_debug_copy_to_device(self.ptr, host.ctypes.data, self.nbytes) == 0

    def numpy(self) -> np.ndarray:
        assert hip().hipDeviceSynchronize() == 0
        out = np.empty(self.shape, self.dtype)
        assert _psm().psm_debug_copy_to_host(out.ctypes.data, self.ptr, self.nbytes) == 0
        return out

    def free(self):
        if self.ptr:
            if self._guarded:
                self._lib.psm_debug_free(self.ptr)
            else:
                hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
