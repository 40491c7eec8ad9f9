"""The MPI funnel of the drop-in module (python_module.py:179-185 gather x4 in init_func, :258 gather and :501-511
slice + scatter in py_func) executed with a fake `mpi4py`: four in-process "ranks" (threads, one module instance each)
with the uneven 4-way split of a decomposed case (system/decomposeParDict: numberOfSubdomains 4), some ranks owning no
boundary faces.  On the CPU the rank-0 solver object is a recording stub; on the GPU it is the real SolverModule and
every rank's slice must equal the serial call."""
import importlib.util
import os
import sys
import threading
import types

import numpy as np
import pytest

import cases
from test_python_module import case_dir  # noqa: F401  (fixture: artefact files in the working directory)
from psm_amd import SolverModule

PKG = os.path.dirname(importlib.util.find_spec("psm_amd.python_module").origin)
NPROCS = 4


class _World:
    def __init__(self, n):
        self.n, self.slots, self.box = n, [None] * n, None
        self.barrier = threading.Barrier(n)
        self.local = threading.local()


class _Comm:
    """Object-mode gather / scatter like mpi4py's lowercase methods (lists in rank order on the root)."""

    def __init__(self, world):
        self.w = world

    def Get_rank(self):
        return self.w.local.rank

    def Get_size(self):
        return self.w.n

    def gather(self, obj, root=0):
        r = self.Get_rank()
        self.w.slots[r] = obj
        self.w.barrier.wait()
        out = list(self.w.slots) if r == root else None
        self.w.barrier.wait()
        return out

    def scatter(self, objs, root=0):
        r = self.Get_rank()
        if r == root:
            assert len(objs) == self.w.n
            self.w.box = objs
        self.w.barrier.wait()
        out = self.w.box[r]
        self.w.barrier.wait()
        return out


def _install_fake_mpi(monkeypatch, world, initialized=True):
    comm = _Comm(world)
    mpi = types.ModuleType("mpi4py.MPI")
    mpi.COMM_WORLD = comm
    mpi.Is_initialized = lambda: initialized
    pkg = types.ModuleType("mpi4py")
    pkg.rc = types.SimpleNamespace(initialize=True, finalize=True)
    pkg.MPI = mpi
    monkeypatch.setitem(sys.modules, "mpi4py", pkg)
    monkeypatch.setitem(sys.modules, "mpi4py.MPI", mpi)
    return pkg


def _load_rank_module(r):
    spec = importlib.util.spec_from_file_location(f"psm_python_module_rank{r}", os.path.join(PKG, "python_module.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _split(array, top, obst):
    """Uneven contiguous 4-way split; ranks 1 and 3 own no obstacle faces, rank 2 no `top` faces."""
    n = array.shape[0]
    cuts = [0, int(0.31 * n), int(0.48 * n), int(0.80 * n), n]
    arrays = [array[cuts[r]:cuts[r + 1]] for r in range(NPROCS)]
    tcut = [0, len(top) // 3, len(top) // 3 * 2, len(top) // 3 * 2, len(top)]
    tops = [top[tcut[r]:tcut[r + 1]] for r in range(NPROCS)]
    ocut = [0, len(obst) // 2, len(obst) // 2, len(obst), len(obst)]
    obsts = [obst[ocut[r]:ocut[r + 1]] for r in range(NPROCS)]
    return arrays, tops, obsts


def _run_ranks(world, body):
    errs, outs = [], [None] * world.n

    def work(r):
        world.local.rank = r
        try:
            outs[r] = body(r)
        except BaseException as e:             # a failed rank must not leave the others in a barrier
            errs.append((r, e))
            world.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world.n)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs
    return outs


class _StubSolver:
    def __init__(self):
        self.init_args, self.calls = None, 0

    def init_func(self, array, top, obst):
        self.init_args = (np.array(array), np.array(top), np.array(obst))
        return 0

    def pin(self, array, out=None):                     # the module registers its persistent gather buffer (rank 0, parallel solver)
        self.pinned = (array, out)

    def unpin(self):
        self.pinned = None

    def py_func(self, array, out=None):
        self.calls += 1
        p = 2.0 * array[:, 0] - array[:, 3] + 0.5 * array[:, 4] + self.calls
        if out is None:
            return p
        assert self.pinned is not None and array is self.pinned[0] and out is self.pinned[1]     # the registered pair, every step
        out[:] = p
        return out


def test_gather_slice_scatter_with_four_fake_ranks(case_dir, monkeypatch):  # noqa: F811
    array, top, obst, model, maxs = case_dir
    world = _World(NPROCS)
    pkg = _install_fake_mpi(monkeypatch, world)
    arrays, tops, obsts = _split(array, top, obst)
    stub = _StubSolver()
    mods = [None] * NPROCS

    def body(r):
        pm = _load_rank_module(r)
        mods[r] = pm
        assert (pm.rank, pm.nprocs) == (r, NPROCS) and pm.comm is not None
        assert (pm._module is not None) == (r == 0)                     # only rank 0 holds the model (python_module.py:168-170)
        if r == 0:
            pm._module = stub
        assert pm.init_func(arrays[r], tops[r], obsts[r], r) == 0
        p1 = pm.py_func(arrays[r], r)
        p2 = pm.py_func(arrays[r] * 1.5, r)
        return p1, p2
    outs = _run_ranks(world, body)
    assert pkg.rc.initialize is False and pkg.rc.finalize is False     # MPI_Init / Finalize stay with the solver
    np.testing.assert_array_equal(stub.init_args[0], array)            # rank order == concatenation order
    np.testing.assert_array_equal(stub.init_args[1], top)
    np.testing.assert_array_equal(stub.init_args[2], obst)
    assert mods[0].len_rankwise == [a.shape[0] for a in arrays] and mods[1].len_rankwise is None
    ref1 = 2.0 * array[:, 0] - array[:, 3] + 0.5 * array[:, 4] + 1
    a15 = array * 1.5
    ref2 = 2.0 * a15[:, 0] - a15[:, 3] + 0.5 * a15[:, 4] + 2
    start = 0
    for r in range(NPROCS):
        n = arrays[r].shape[0]
        np.testing.assert_array_equal(outs[r][0], ref1[start:start + n])
        np.testing.assert_array_equal(outs[r][1], ref2[start:start + n])
        start += n
    assert start == array.shape[0]


def test_uninitialised_mpi_is_left_alone(case_dir, monkeypatch):  # noqa: F811
    """Serial solver on a host that has mpi4py: MPI_Init never ran, COMM_WORLD must not be touched."""
    world = _World(1)
    pkg = _install_fake_mpi(monkeypatch, world, initialized=False)

    class Boom:
        def __getattr__(self, name):
            raise AssertionError("COMM_WORLD used before MPI_Init")
    pkg.MPI.COMM_WORLD = Boom()
    pm = _load_rank_module(0)
    assert pm.comm is None and (pm.rank, pm.nprocs) == (0, 1) and pm._module is not None


@pytest.mark.gpu
def test_four_fake_ranks_equal_the_serial_call(case_dir, monkeypatch):  # noqa: F811
    array, top, obst, model, maxs = case_dir
    world = _World(NPROCS)
    _install_fake_mpi(monkeypatch, world)
    arrays, tops, obsts = _split(array, top, obst)

    def body(r):
        pm = _load_rank_module(r)
        assert pm.init_func(arrays[r], tops[r], obsts[r], r) == 0
        return pm.py_func(arrays[r], r)
    outs = _run_ranks(world, body)
    ref = SolverModule(model, maxs)
    ref.init_func(array, top, obst)
    np.testing.assert_array_equal(np.concatenate(outs), ref.py_func(array))
