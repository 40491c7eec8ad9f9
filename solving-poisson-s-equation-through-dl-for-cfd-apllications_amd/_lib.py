"""ctypes binding of the C-ABI in include/psm.h (libpsm_hip.so, built in-tree by
``__graft_entry__.build()`` / ``make -C csrc``).

The product path has no fallback: if the shared library is missing or cannot be
loaded this module raises ``PsmLibraryError`` -- it never routes to NumPy or to
the test oracle.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSM_LIB") or os.path.join(HERE, "libpsm_hip.so")   # PSM_LIB: diagnostic builds only
HEADER = os.path.join(os.path.dirname(HERE), "include", "psm.h")
HEADER_UNET = os.path.join(os.path.dirname(HERE), "include", "psm_unet.h")

PSM_ABI_VERSION = 4
VARIANTS = {"chapter5": 0, "deltas": 1, "gradp": 2}
SCALERS = {"max_abs": 0, "std": 1, "min_max": 2}
PRECISIONS = {"f32": 0, "bf16": 1}
STAGES = {"x_input": 0, "res": 1, "block_pred": 2, "offsets": 3, "shift": 4}
KERNELS = ("encode", "reduce", "mlp", "decode", "strips", "chain", "paste")
ERRORS = {0: "PSM_OK", -1: "PSM_ERR_ARG", -2: "PSM_ERR_STATE", -3: "PSM_ERR_HIP", -4: "PSM_ERR_NO_DEVICE",
          -5: "PSM_ERR_UNSUPPORTED", -6: "PSM_ERR_NOMEM", -7: "PSM_ERR_GEOMETRY"}


class PsmLibraryError(RuntimeError):
    pass


class PsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class psm_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "variant", "block", "overlap", "c_in", "c_out", "p_in", "p_out", "n_dense", "scaler",
        "sdf_channel", "device", "max_cases", "strict_degenerate", "precision")]


_f32p, _f64p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
_hp = C.c_void_p

# name -> (restype, argtypes): every symbol include/psm.h declares
SIGNATURES = {
    "psm_create": (C.c_int, [C.POINTER(psm_config), C.POINTER(_hp)]),
    "psm_destroy": (None, [_hp]),
    "psm_last_error": (C.c_char_p, [_hp]),
    "psm_set_pca": (C.c_int, [_hp, _f64p, _f64p, _f64p, _f64p]),
    "psm_set_dense": (C.c_int, [_hp, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p]),
    "psm_set_attention": (C.c_int, [_hp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p, _f32p, _f32p]),
    "psm_set_layernorm": (C.c_int, [_hp, C.c_int32, C.c_int32, _f32p, _f32p, C.c_float, C.c_int32]),
    "psm_set_conv1d": (C.c_int, [_hp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p]),
    "psm_set_scaler": (C.c_int, [_hp, _f64p, _f64p, _f64p, _f64p]),
    "psm_plan_grid": (C.c_int, [_hp, C.c_int32, C.c_int32]),
    "psm_num_blocks": (C.c_int, [_hp]),
    "psm_bind_geometry": (C.c_int, [_hp, C.c_void_p, C.c_int32]),
    "psm_bind_geometry_cases": (C.c_int, [_hp, C.c_void_p, C.c_int32, C.c_int32]),
    "psm_unbind_geometry": (C.c_int, [_hp]),
    "psm_geometry_bound": (C.c_int, [_hp]),
    "psm_bound_mask": (C.c_int, [_hp, C.POINTER(C.c_uint8), C.c_size_t]),
    "psm_guard_trips": (C.c_int64, [_hp]),
    "psm_solve_grid": (C.c_int, [_hp, _f32p, C.c_int32, _f32p, _f32p]),
    "psm_grid_shape": (C.c_int, [_hp, _i32p]),
    "psm_ring_acquire": (C.c_int, [_hp, C.POINTER(C.c_int64), C.POINTER(_f32p), C.POINTER(_f32p)]),
    "psm_ring_submit": (C.c_int, [_hp, C.c_int64, C.c_int32, _f32p]),
    "psm_ring_wait": (C.c_int, [_hp, C.c_int64]),
    "psm_ring_release": (C.c_int, [_hp, C.c_int64]),
    "psm_host_register": (C.c_int, [_hp, C.c_void_p, C.c_size_t]),
    "psm_host_unregister": (C.c_int, [_hp, C.c_void_p]),
    "psm_submit_grid_io": (C.c_int, [_hp, _f32p, C.c_int32, _f32p, _f32p, C.POINTER(C.c_int64)]),
    "psm_submit_grid": (C.c_int, [_hp, _f32p, C.c_int32, _f32p, C.POINTER(C.c_int64)]),
    "psm_wait_grid": (C.c_int, [_hp, C.c_int64, _f32p]),
    "psm_solve_grid_device": (C.c_int, [_hp, C.c_void_p, C.c_int32, _f32p, C.c_void_p, C.c_void_p]),
    "psm_reassemble": (C.c_int, [_hp, _f32p, _f32p, _f32p]),
    "psm_label_blocks": (C.c_int, [_hp, _f32p, _f32p, _f32p]),
    "psm_block_error": (C.c_int, [_hp, _f32p, _f32p, C.POINTER(C.c_double)]),
    "psm_set_geometry": (C.c_int, [_hp, C.c_int64, C.c_int32, C.c_int32, _i32p, _f64p, _i32p, _f64p, _i32p, _f64p, _f64p,
                                   C.c_int32, C.c_int32, C.c_double]),
    "psm_set_case": (C.c_int, [_hp, _f64p, C.c_double, C.c_int32, C.c_double]),
    "psm_init_geometry": (C.c_int, [_hp, _f64p, C.c_int64, _f64p, C.c_int64, _f64p, C.c_int64, C.c_int32]),
    "psm_geometry_shape": (C.c_int, [_f64p, C.c_int64, C.c_double, _i32p, _i32p, _f64p]),
    "psm_geometry_build": (C.c_int, [_f64p, C.c_int64, _f64p, C.c_int64, _f64p, C.c_int64, C.c_double, C.c_int32, _i32p, _f64p,
                                     _i32p, _f64p, _i32p, _f64p]),
    "psm_geometry_last_error": (C.c_char_p, []),
    "psm_solve": (C.c_int, [_hp, _f64p, C.c_int64, C.c_int32, _f64p]),
    "psm_solve_begin": (C.c_int, [_hp, _f64p, C.c_int64, C.c_int32, _f64p]),
    "psm_solve_end": (C.c_int, [_hp]),
    "psm_poisson_features": (C.c_int, [_hp, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_int32, C.c_int32, _f64p, _f32p]),
    "psm_pin_buffers": (C.c_int, [_hp, _f64p, _f64p]),
    "psm_unpin_buffers": (C.c_int, [_hp]),
    "psm_mesh_to_grid": (C.c_int, [_hp, _f64p, C.c_int64, C.c_int32, C.c_int32, _f64p]),
    "psm_gaussian_filter": (C.c_int, [_hp, _f32p, C.c_int32, C.c_int32, C.c_double, C.c_double, _f32p]),
    "psm_set_integration": (C.c_int, [_hp, C.c_int32, C.c_int32, _f64p, C.c_int32, C.c_int32, C.c_double, C.c_double]),
    "psm_integrate_gradp": (C.c_int, [_hp, _f32p, _f32p]),
    "psm_synchronize": (C.c_int, [_hp]),
    "psm_read_stage": (C.c_int, [_hp, C.c_int32, _f32p, C.c_size_t]),
    "psm_profile_solve": (C.c_int, [_hp, C.c_void_p, C.c_int32, C.c_void_p, _f32p]),
    "psm_enable_kernel_timing": (C.c_int, [_hp, C.c_int32, C.c_int32]),
    "psm_get_kernel_timing": (C.c_int, [_hp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "psm_event_pair_overhead": (C.c_int, [_hp, C.c_int32, C.POINTER(C.c_double)]),
    "psm_time_kernels": (C.c_int, [_hp, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64), C.c_int32, _i32p]),
    "psm_time_kernels_q": (C.c_int, [_hp, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32, _i32p]),
    "psm_bench_host": (C.c_int, [_hp, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(C.c_double), _f32p]),
    "psm_layout": (C.c_int, [C.c_int32] * 5 + [_i32p, C.c_int32, _i32p, _i32p]),
    "psm_owner_map": (C.c_int, [C.c_int32] * 6 + [_i32p]),
    "psm_debug_reassemble_host": (C.c_int, [C.c_int32] * 9 + [_f32p, _f32p, _f32p, _f32p, _f32p]),
    "psm_debug_guard_pages": (C.c_int, []),
    "psm_debug_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "psm_debug_free": (C.c_int, [C.c_void_p]),
    "psm_debug_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "psm_debug_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "psm_abi_version": (C.c_int, []),
}

# include/psm_unet.h (convolutional path)
_up = C.c_void_p
_i32ptr = C.POINTER(C.c_int32)
SIGNATURES_UNET = {
    "psm_unet_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _i32ptr, C.c_int32, C.POINTER(_up)]),
    "psm_unet_destroy": (None, [_up]),
    "psm_unet_last_error": (C.c_char_p, [_up]),
    "psm_unet_num_convs": (C.c_int, [_up]),
    "psm_unet_conv_shape": (C.c_int, [_up, C.c_int32, _i32ptr, _i32ptr, _i32ptr]),
    "psm_unet_set_conv": (C.c_int, [_up, C.c_int32, _f32p, _f32p]),
    "psm_unet_set_precision": (C.c_int, [_up, C.c_int32]),
    "psm_unet_keep_activations": (C.c_int, [_up, C.c_int32]),
    "psm_unet_plan": (C.c_int, [_up, C.c_int32, C.c_int32, C.c_int32]),
    "psm_unet_forward": (C.c_int, [_up, _f32p, C.c_int32, _f32p]),
    "psm_unet_forward_device": (C.c_int, [_up, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "psm_unet_synchronize": (C.c_int, [_up]),
    "psm_unet_read_activation": (C.c_int, [_up, C.c_int32, _f32p, C.c_int64]),
    "psm_unet_profile": (C.c_int, [_up, C.c_void_p, C.c_int32, C.c_void_p, _f32p, _i32ptr]),
    "psm_unet_autotune": (C.c_int, [_up, C.c_int32, C.c_int32, _f32p, _f32p]),
    "psm_unet_get_choices": (C.c_int, [_up, _i32ptr, C.c_int32]),
    "psm_unet_set_choices": (C.c_int, [_up, _i32ptr, C.c_int32]),
    "psm_unet_ksplit": (C.c_int, [_up, C.c_int32]),
    "psm_unet_plan_info": (C.c_int, [_up, C.c_int32, _i32ptr]),
    "psm_unet_time_kernels": (C.c_int, [_up, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_double), _i32ptr, C.c_char_p]),
    "psm_unet_time_kernels_q": (C.c_int, [_up, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), _i32ptr, C.c_char_p]),
    "psm_unet_debug_run_layer": (C.c_int, [_up, C.c_int32, _f32p]),
    "psm_unet_flops": (C.c_int64, [_up]),
}

_lib = None


def load():
    """Load libpsm_hip.so (once) and bind every symbol; raises PsmLibraryError."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsmLibraryError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
            "there is no CPU fallback for the surrogate path")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise PsmLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in list(SIGNATURES.items()) + list(SIGNATURES_UNET.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PsmLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.psm_abi_version() != PSM_ABI_VERSION:
        raise PsmLibraryError("libpsm_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().psm_last_error(handle)
    return msg.decode() if msg else ""


def check(rc, handle=None):
    if rc != 0:
        raise PsmError(rc, last_error(handle))
    return rc
