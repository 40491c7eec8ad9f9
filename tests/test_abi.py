"""The C-ABI library loads and exports every symbol include/psm.h declares;
without a GPU the product path fails loudly instead of falling back."""
import ctypes as C
import re

import pytest

from psm_amd import _lib


def declared_functions(header=None):
    txt = open(header or _lib.HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psm_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert names, "no declarations found"
    assert sorted(_lib.SIGNATURES) == names


def test_unet_header_and_binding_agree():
    names = declared_functions(_lib.HEADER_UNET)
    assert names and sorted(_lib.SIGNATURES_UNET) == names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_functions() + declared_functions(_lib.HEADER_UNET):
        assert hasattr(lib, name), name
    assert lib.psm_abi_version() == _lib.PSM_ABI_VERSION


def test_create_rejects_bad_config_and_reports():
    lib = _lib.load()
    cfg = _lib.psm_config(abi_version=_lib.PSM_ABI_VERSION, variant=7, block=128, c_in=3, c_out=1, p_in=8, p_out=8,
                          n_dense=2, max_cases=1)
    h = C.c_void_p()
    rc = lib.psm_create(C.byref(cfg), C.byref(h))
    assert rc == -1 and not h.value
    assert "variant" in _lib.last_error()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = _lib.psm_config(abi_version=_lib.PSM_ABI_VERSION, variant=1, block=128, c_in=3, c_out=1, p_in=8, p_out=8,
                          n_dense=2, sdf_channel=2, max_cases=1)
    h = C.c_void_p()
    rc = lib.psm_create(C.byref(cfg), C.byref(h))
    assert rc == -4 and not h.value          # PSM_ERR_NO_DEVICE
    assert "no CPU fallback" in _lib.last_error()


def test_headers_compile_as_c99():
    """include/psm.h and include/psm_unet.h are plain C (the boundary a cgo / JNI / Fortran binding would use)."""
    import os
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c",
                            os.path.join(root, "tests", "native", "abi_c_check.c"), "-o", os.path.join(td, "a.o")],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
