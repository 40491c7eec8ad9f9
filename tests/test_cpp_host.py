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


def _build(out, src="ensemble_host.cpp", extra=()):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src), *extra,
           "-L", PKG, "-lpsm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True)


RCCL_FLAGS = ("-DPSM_WITH_RCCL", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-lrccl", "-lamdhip64")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpp_host_builds_against_the_c_abi(tmp_path):
    r = _build(str(tmp_path / "ensemble_host"))
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_multi_device_host_builds_with_and_without_rccl(tmp_path):
    """examples/ensemble_multi.cpp: one handle + one thread per device, contiguous case shard; the RCCL variant (model payload
    broadcast with ncclBroadcast from C++, no torch) links against librccl / libamdhip64 of the image."""
    r = _build(str(tmp_path / "ensemble_multi"), "ensemble_multi.cpp")
    assert r.returncode == 0, r.stderr[-2000:]
    if os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        r = _build(str(tmp_path / "ensemble_multi_rccl"), "ensemble_multi.cpp", RCCL_FLAGS)
        assert r.returncode == 0, r.stderr[-2000:]


def test_cpp_shard_rule_is_dist_shard_cases():
    """The C++ host partitions the case batch with the rule of dist.shard_cases (read from the source: the test has no GPU)."""
    from psm_amd import dist as pdist
    src = open(os.path.join(ROOT, "examples", "ensemble_multi.cpp")).read()
    assert "*first = rank * base + (rank < extra ? rank : extra);" in src and "*count = base + (rank < extra ? 1 : 0);" in src
    for n, w in ((64, 8), (7, 2), (5, 8), (0, 3)):
        got = []
        for r in range(w):
            base, extra = divmod(n, w)
            got.append((r * base + min(r, extra), base + (1 if r < extra else 0)))
        assert got == [pdist.shard_cases(n, w, r) for r in range(w)]
        assert sum(c for _, c in got) == n and all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(w - 1))


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


@pytest.mark.gpu
def test_multi_device_host_shards_cases_over_handles(tmp_path):
    """examples/ensemble_multi.cpp on the one-GPU box: one device slot, then TWO handles (threads) on device 0 -- the multi-device
    code path with both shards on one card.  One case per call: every case takes the single-case launch sequence, so the fields
    equal the Python mirror's single solves bit for bit whatever the sharding; case batches of four per call: float32 summation
    order of the batch path.  The RCCL build broadcasts the model payload (one device: the degenerate collective) and gives the
    same fields."""
    exe = str(tmp_path / "ensemble_multi")
    r = _build(exe, "ensemble_multi.cpp")
    assert r.returncode == 0, r.stderr[-2000:]
    model = synthetic.make_model("deltas", p_in=32, p_out=32, seed_pca=77, seed_w=8)
    n = 7
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=5).astype(np.float32)
    _write_model(tmp_path / "model.bin", model, 256, 256)
    grids.tofile(tmp_path / "grids.bin")
    with GridSurrogate(model, 256, 256) as sur:
        ref = np.stack([sur.solve(grids[k])[0] for k in range(n)])

    def run(exe_, *args):
        out = tmp_path / "fields.bin"
        if out.exists():
            out.unlink()
        p = subprocess.run([exe_, str(tmp_path / "model.bin"), str(tmp_path / "grids.bin"), str(n), str(out), *args], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (p.stdout, p.stderr)
        return np.fromfile(out, np.float32).reshape(n, 256, 256, 1), p.stdout

    one, txt = run(exe, "--devices", "0", "--batch", "1")
    assert "cases [0, 7)" in txt and "solves/s" in txt
    np.testing.assert_array_equal(one, ref)
    two, txt = run(exe, "--devices", "0,0", "--batch", "1")
    assert "cases [0, 4)" in txt and "cases [4, 7)" in txt                      # dist.shard_cases(7, 2, r)
    np.testing.assert_array_equal(two, ref)
    three, txt = run(exe, "--devices", "0,0,0", "--batch", "4", "--repeat", "2")
    assert "cases [0, 3)" in txt and "cases [3, 5)" in txt and "cases [5, 7)" in txt
    np.testing.assert_allclose(three, ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    if os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        exe2 = str(tmp_path / "ensemble_multi_rccl")
        r = _build(exe2, "ensemble_multi.cpp", RCCL_FLAGS)
        assert r.returncode == 0, r.stderr[-2000:]
        got, txt = run(exe2, "--devices", "0", "--batch", "1", "--rccl-broadcast")
        assert "broadcast from device 0" in txt
        np.testing.assert_array_equal(got, ref)
        bad = subprocess.run([exe2, str(tmp_path / "model.bin"), str(tmp_path / "grids.bin"), str(n), str(tmp_path / "x.bin"), "--devices", "0,0", "--rccl-broadcast"],
                             capture_output=True, text=True, timeout=120)
        assert bad.returncode != 0 and "distinct devices" in bad.stderr
