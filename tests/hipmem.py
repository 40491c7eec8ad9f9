"""Minimal HIP runtime access through ctypes for the GPU tests (device buffers without importing
torch, whose first import on a fresh box can take minutes).  With PSM_GUARD_PAGES=1 in the environment the buffers come
from the library's guard-page allocator (csrc/psm_alloc.cpp): a kernel that runs past the end of a test's input or
output buffer faults instead of touching a neighbour."""
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
            from psm_amd import _lib
            lib = _lib.load()
            assert lib.psm_debug_guard_pages() == 1 and lib.psm_debug_malloc(C.byref(p), self.nbytes) == 0
            self._lib = lib
        else:
            assert hip().hipMalloc(C.byref(p), self.nbytes) == 0
        self.ptr = p.value
        if host is not None:
            assert hip().hipMemcpy(self.ptr, host.ctypes.data, self.nbytes, 1) == 0      # H2D

    def numpy(self) -> np.ndarray:
        assert hip().hipDeviceSynchronize() == 0
        out = np.empty(self.shape, self.dtype)
        assert hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0           # D2H
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
