"""A C++ host (examples/ensemble_host.cpp) built against include/psm.h + libpsm_hip.so only: compiles here with g++
(CPU check), and on the GPU drives an ensemble of cases through the submit / wait ring with the same fields as the
Python mirror."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from psm_amd import GridSurrogate, _lib, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(_lib.LIB_PATH)


def _build(out):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ensemble_host.cpp"),
           "-L", PKG, "-lpsm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpp_host_builds_against_the_c_abi(tmp_path):
    r = _build(str(tmp_path / "ensemble_host"))
    assert r.returncode == 0, r.stderr[-2000:]


def _write_model(path, model, ny, nx):
    sc = {"max_abs": 0, "std": 1, "min_max": 2}[model.scaler_kind]
    with open(path, "wb") as f:
        f.write(struct.pack("<9i", {"chapter5": 0, "deltas": 1, "gradp": 2}[model.variant], model.c_in, model.c_out, model.p_in,
                            model.p_out, len(model.weights), sc, ny, nx))
        for a in (model.comp_in, model.mean_in, model.comp_out, model.mean_out):
            f.write(np.ascontiguousarray(a, "<f8").tobytes())
        n_in, n_out = (1, 1) if sc == 0 else (model.p_in, model.p_out)
        for a, n in ((model.in_a, n_in), (model.in_b, n_in), (model.out_a, n_out), (model.out_b, n_out)):
            f.write(np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), (n,)), "<f8").tobytes())
        for W, b in model.weights:
            f.write(struct.pack("<2i", *W.shape))
            f.write(np.ascontiguousarray(W, "<f4").tobytes()); f.write(np.ascontiguousarray(b, "<f4").tobytes())


@pytest.mark.gpu
def test_cpp_host_matches_python_mirror(tmp_path):
    exe = str(tmp_path / "ensemble_host")
    r = _build(exe)
    assert r.returncode == 0, r.stderr[-2000:]
    model = synthetic.make_model("deltas", p_in=32, p_out=32, seed_pca=77, seed_w=8)
    n = 7
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=5).astype(np.float32)
    _write_model(tmp_path / "model.bin", model, 256, 256)
    grids.tofile(tmp_path / "grids.bin")
    run = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "grids.bin"), str(n), str(tmp_path / "fields.bin")],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr)
    assert "solves/s" in run.stdout
    got = np.fromfile(tmp_path / "fields.bin", np.float32).reshape(n, 256, 256, 1)
    with GridSurrogate(model, 256, 256) as sur:
        for k in range(n):
            np.testing.assert_array_equal(got[k], sur.solve(grids[k])[0])
