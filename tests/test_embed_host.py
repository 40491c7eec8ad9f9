"""The reference's REAL binding, run for once: tests/native/embed_host.cpp restates the CPython C-API sequence of
DLPoissonFoam (PythonComm_init.H:3-21,53-55,80-94; PythonComm.H:3-36; serial form: singleCore/DLPoissonSolver_1) and drives
whatever `python_module.py` sits in its working directory -- here the drop-in shim, copied into a temporary case directory
next to the artefact files the reference reads (python_module.py:103-118,170).

CPU: the host builds, drives a NumPy-only python_module through 60 steps with one reused argument tuple, and reports an import
failure with exit code 3 instead of the reference's segfault (test_case/log.DL:34-42).
GPU: serial (3 / 1 arguments) and parallel (4 / 2 arguments) binding, default and PSM_PIN_SOLVER_BUFFERS=1: pressures of every
step bit-identical to SolverModule.py_func, reference counts of the shim's persistent arrays flat over the run."""
import os
import re
import shutil
import subprocess
import sysconfig

import numpy as np
import pytest

import cases
from test_python_module import case_dir  # noqa: F401  (fixture: artefact files + cwd + PSM_AMD_HOME)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "solving-poisson-s-equation-through-dl-for-cfd-apllications_amd", "python_module.py")
STEPS = 60


def _toolchain():
    inc = sysconfig.get_paths()["include"]
    libdir = sysconfig.get_config_var("LIBDIR") or ""
    ver = sysconfig.get_config_var("LDVERSION") or sysconfig.get_python_version()
    have = shutil.which("g++") and os.path.exists(os.path.join(inc, "Python.h")) and os.path.exists(os.path.join(np.get_include(), "numpy", "arrayobject.h"))
    so = any(os.path.exists(os.path.join(d, f"libpython{ver}.so")) for d in (libdir, "/usr/lib/x86_64-linux-gnu") if d)
    return (inc, libdir, ver) if have and so else None


needs_embed = pytest.mark.skipif(_toolchain() is None, reason="needs g++, Python.h, libpython and NumPy's C headers")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    inc, libdir, ver = _toolchain()
    exe = str(tmp_path_factory.mktemp("embed") / "embed_host")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "native", "embed_host.cpp"), "-I", inc, "-I", np.get_include(),
           "-L", libdir, f"-lpython{ver}", "-ldl", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _write_mesh(d, array, top, obst):
    for name, a in (("cells.f64", array), ("top.f64", top), ("obst.f64", obst)):
        np.ascontiguousarray(a, "<f8").tofile(os.path.join(d, name))


def _cmd(host, mode, steps, tag="", rank=None):
    return [host, mode, str(steps), f"cells{tag}.f64", f"top{tag}.f64", f"obst{tag}.f64", f"out{tag}.f64"] + ([str(rank)] if rank is not None else [])


def _env(env=None):
    e = dict(os.environ, **(env or {}))
    e.pop("PYTHONPATH", None)                  # the solver's environment: only "." is appended (PythonComm_init.H:5)
    return e


def _run(host, d, mode, steps=STEPS, env=None):
    return subprocess.run(_cmd(host, mode, steps), cwd=d, env=_env(env), capture_output=True, text=True, timeout=600)


def _refcounts(stdout):
    rows = {}
    for m in re.finditer(r"REFCNT step (\d+) (.*)", stdout):
        kv = m.group(2).split()
        rows[int(m.group(1))] = {kv[i]: int(kv[i + 1]) for i in range(0, len(kv), 2)}
    return rows


def _replay(py_func, array, steps):
    """The host's time loop (embed_host.cpp): U scaled per step, column 4 = the pressure of the previous step."""
    a = np.ascontiguousarray(array, np.float64).copy()
    p = a[:, 4].copy()
    out = []
    for s in range(steps):
        scale = 1.0 + 0.01 * float(s % 7)
        a[:, 0], a[:, 1] = array[:, 0] * scale, array[:, 1] * scale
        a[:, 4] = p
        p = np.array(py_func(a), np.float64).reshape(a.shape[0], -1)[:, 0].copy()
        out.append(p)
    return np.stack(out)


NUMPY_MODULE = '''
import numpy as np
calls = {"init": [], "py": 0, "ids": set()}
_pin_state = {"array": None, "out": None}
_cat = {"buf": None, "out": None}
def init_func(array, top, obst, placeholder=None):
    calls["init"].append((array.shape, top.shape, obst.shape, placeholder))
    assert array.flags.c_contiguous and array.dtype == np.float64 and array.base is None
    return 0
def py_func(array, placeholder=None):
    calls["py"] += 1
    calls["ids"].add(array.ctypes.data)
    assert len(calls["ids"]) == 1, "the solver hands over ONE persistent buffer (PythonComm_init.H:53)"
    assert (placeholder is None) == (calls["init"][0][3] is None)      # serial: no rank argument in either call
    if _pin_state["array"] is None:
        _pin_state["array"], _pin_state["out"] = array, np.empty(array.shape[0])
    np.add(0.5 * array[:, 4], array[:, 0] - 0.25 * array[:, 1], out=_pin_state["out"])
    return _pin_state["out"]
'''


@needs_embed
@pytest.mark.parametrize("mode", ["serial", "parallel"])
def test_host_drives_a_numpy_module_through_the_reference_sequence(host, tmp_path, mode):
    """CPU check of the host itself: one py_args tuple reused for 60 steps, a fresh view over the same memory per step, the result
    read through PyArray_GETPTR2 from a 1-D array (what the reference's py_func returns, python_module.py:491-517)."""
    array, top, obst, _, _ = cases.build_mesh_case()
    _write_mesh(tmp_path, array, top, obst)
    (tmp_path / "python_module.py").write_text(NUMPY_MODULE)
    r = _run(host, tmp_path, mode)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.fromfile(tmp_path / "out.f64", "<f8").reshape(STEPS, -1)
    ref = _replay(lambda a: 0.5 * a[:, 4] + (a[:, 0] - 0.25 * a[:, 1]), array, STEPS)
    np.testing.assert_array_equal(got, ref)
    rc = _refcounts(r.stdout)
    assert set(rc) == {2, STEPS - 1}
    # the module's persistent view: module dict only, every step (the tuple's view of each later step is released by the next SetItem)
    assert rc[2]["pin_array"] == rc[STEPS - 1]["pin_array"]
    # the solver never releases pValue (PythonComm.H:24-36): a module that returns ONE persistent array sees exactly one more
    # reference per step -- and no new allocation
    assert rc[STEPS - 1]["pin_out"] - rc[2]["pin_out"] == STEPS - 3
    assert rc[2]["py_args"] == rc[STEPS - 1]["py_args"] == 1


@needs_embed
def test_import_failure_is_an_exit_code_not_a_segfault(host, tmp_path):
    """No python_module.py in the case directory / a module whose import raises: the reference dereferences NULL in
    PyObject_GetAttrString (test_case/log.DL:34-42); the host reports the traceback and exits with 3."""
    array, top, obst, _, _ = cases.build_mesh_case()
    _write_mesh(tmp_path, array, top, obst)
    r = _run(host, tmp_path, "parallel", steps=2)
    assert r.returncode == 3 and "ModuleNotFoundError" in r.stderr and "import python_module failed" in r.stderr
    shutil.copy(SHIM, tmp_path / "python_module.py")               # the shim without its artefact files: OSError on import
    r = _run(host, tmp_path, "parallel", steps=2, env={"PSM_AMD_HOME": ROOT})
    assert r.returncode == 3 and "import python_module failed" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert "Error" in r.stderr
    (tmp_path / "python_module.py").write_text("def init_func(*a):\n    return 0\n")
    r = _run(host, tmp_path, "parallel", steps=2)
    assert r.returncode == 3 and "lacks py_func" in r.stderr


@needs_embed
@pytest.mark.gpu
@pytest.mark.parametrize("mode,pin", [("serial", "0"), ("serial", "1"), ("parallel", "0"), ("parallel", "1")])
def test_embedded_solver_sequence_equals_solver_module(host, case_dir, mode, pin):  # noqa: F811
    from psm_amd import SolverModule
    array, top, obst, model, maxs = case_dir
    d = os.getcwd()                                                 # the fixture's case directory (artefact files written there)
    shutil.copy(SHIM, os.path.join(d, "python_module.py"))          # "copy this file into the case directory as python_module.py"
    _write_mesh(d, array, top, obst)
    r = _run(host, d, mode, env={"PSM_AMD_HOME": ROOT, "PSM_PIN_SOLVER_BUFFERS": pin})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Traceback" not in r.stderr, r.stderr[-3000:]            # py_func swallows GPU-side failures and returns the previous p: none may occur
    got = np.fromfile(os.path.join(d, "out.f64"), "<f8").reshape(STEPS, -1)
    ref_mod = SolverModule(model, maxs)
    ref_mod.init_func(array, top, obst)
    ref = _replay(ref_mod.py_func, array, STEPS)
    np.testing.assert_array_equal(got, ref)
    assert np.isfinite(got).all() and np.abs(got[-1] - got[0]).max() > 0     # the steps differ: the loop really fed p back
    rc = _refcounts(r.stdout)
    a, b = rc[2], rc[STEPS - 1]
    assert a["py_args"] == b["py_args"] == 1
    if pin == "1":                                                  # one-rank run, serial or parallel binding: the solver's buffer is registered once
        assert a["pin_array"] > 0 and a["pin_array"] == b["pin_array"]            # the first step's view, held by the module alone
        assert b["pin_out"] - a["pin_out"] == STEPS - 3                           # the caller's one unreleased reference per step, no new array
    else:
        assert a["pin_array"] == b["pin_array"] == -1 and a["pin_out"] == b["pin_out"] == -1
    assert a["cat_buf"] == b["cat_buf"] == -1                       # the gather buffer exists only with more than one rank


# A process-level stand-in for mpi4py (not installable here): object-mode gather / scatter of COMM_WORLD over a Unix socket, rank and
# size from the environment -- what `mpirun -np N DLPoissonFoam -parallel` gives every solver process.  Written into the case
# directory, where the solver's sys.path.append(".") finds it.
FAKE_MPI_INIT = "import types\nrc = types.SimpleNamespace(initialize=True, finalize=True)\n"
FAKE_MPI = '''
import os, time
from multiprocessing.connection import Listener, Client
_rank, _size, _addr = int(os.environ["FAKE_MPI_RANK"]), int(os.environ["FAKE_MPI_SIZE"]), os.environ["FAKE_MPI_ADDR"]
def Is_initialized():
    return True
class _Comm:
    def __init__(self):
        if _rank == 0:
            self.listener = Listener(_addr, family="AF_UNIX")
            self.conns = {}
            for _ in range(_size - 1):
                c = self.listener.accept()
                self.conns[c.recv()] = c
        else:
            for _ in range(1200):
                try:
                    self.c = Client(_addr, family="AF_UNIX")
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    time.sleep(0.1)
            self.c.send(_rank)
    def Get_rank(self):
        return _rank
    def Get_size(self):
        return _size
    def gather(self, obj, root=0):
        if _rank == 0:
            return [obj] + [self.conns[r].recv() for r in range(1, _size)]
        self.c.send(obj)
        return None
    def scatter(self, objs, root=0):
        if _rank == 0:
            for r in range(1, _size):
                self.conns[r].send(objs[r])
            return objs[0]
        return self.c.recv()
COMM_WORLD = _Comm()
'''


@needs_embed
@pytest.mark.gpu
def test_three_embedded_solver_processes_funnel_to_rank_0(host, case_dir):  # noqa: F811
    """The parallel solver as it is deployed: N solver processes, each with its own embedded interpreter, python_module's gather to
    rank 0 / solve / scatter (python_module.py:179-191, 258-264, 501-511).  Three embed_host processes own an uneven split of the
    cells (rank 2 has no boundary faces), only rank 0 touches the GPU; every rank's pressures over 20 steps must be its slice of the
    one-process result, bit for bit."""
    from psm_amd import SolverModule
    array, top, obst, model, maxs = case_dir
    d = os.getcwd()
    shutil.copy(SHIM, os.path.join(d, "python_module.py"))
    os.makedirs(os.path.join(d, "mpi4py"))
    open(os.path.join(d, "mpi4py", "__init__.py"), "w").write(FAKE_MPI_INIT)
    open(os.path.join(d, "mpi4py", "MPI.py"), "w").write(FAKE_MPI)
    n = array.shape[0]
    cuts = [0, n // 2 + 17, n - n // 5, n]                               # uneven three-way decomposition
    tcut = [0, top.shape[0] // 3, top.shape[0], top.shape[0]]            # rank 2: no top faces ...
    ocut = [0, obst.shape[0], obst.shape[0], obst.shape[0]]              # ... ranks 1, 2: no obstacle faces
    for r in range(3):
        for name, a, c in (("cells", array, cuts), ("top", top, tcut), ("obst", obst, ocut)):
            np.ascontiguousarray(a[c[r]:c[r + 1]], "<f8").tofile(os.path.join(d, f"{name}_r{r}.f64"))
    steps = 20
    sock = os.path.join(d, "fake_mpi.sock")
    procs = []
    for r in range(3):
        env = _env({"PSM_AMD_HOME": ROOT, "FAKE_MPI_RANK": str(r), "FAKE_MPI_SIZE": "3", "FAKE_MPI_ADDR": sock})
        procs.append(subprocess.Popen(_cmd(host, "parallel", steps, tag=f"_r{r}", rank=r), cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=600))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and "Traceback" not in se, (r, so[-800:], se[-3000:])
    got = np.concatenate([np.fromfile(os.path.join(d, f"out_r{r}.f64"), "<f8").reshape(steps, -1) for r in range(3)], axis=1)
    ref_mod = SolverModule(model, maxs)
    ref_mod.init_func(array, top, obst)
    ref = _replay(ref_mod.py_func, array, steps)
    np.testing.assert_array_equal(got, ref)
