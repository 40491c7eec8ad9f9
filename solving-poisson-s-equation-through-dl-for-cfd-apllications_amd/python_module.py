"""Drop-in ``python_module`` for the unmodified ``DLPoissonFoam`` solvers (serial and parallel).

The solver embeds CPython, puts "." on ``sys.path`` and fetches the two callables ``init_func`` and
``py_func`` of a module called ``python_module`` (PythonComm_init.H:4-5,11-18); this file keeps those
names, argument lists and the artefact files of
Thesis_Work/Chapter5/{parallelized/test_case,singleCore/test_Case}/python_module.py and routes the work
to the MI355X library (``libpsm_hip.so``) instead of NumPy + TensorFlow.

Use: copy (or symlink) this file into the case directory as ``python_module.py`` and point
``PSM_AMD_HOME`` at the directory that holds ``psm_amd.py`` (the repository root).  Files read from the
working directory, exactly like the reference (python_module.py:103-118,170):
``ipca_input_more.pkl``, ``ipca_p_more.pkl`` (or the ``.npz`` exports), ``maxs``, ``maxs_PCA``, ``weights.h5``.

``PSM_PIN_SOLVER_BUFFERS=1`` (serial solver only, opt-in): the solver hands over the SAME array every step -- ``input_vals`` of
PythonComm_init.H:53 is allocated once and never freed, PythonComm.H:17 wraps it without copying --, so it is registered with
the GPU once (``psm_pin_buffers``) together with one persistent output array, and a step issues no copy at all (65-75 instead of
95 us on a 16 k-cell mesh).  Opt-in because the module cannot see whether the caller's buffer outlives the registration: only a
caller that keeps it allocated for the whole run (the reference solver does) may set it.  The returned array is then the same
object every step (valid until the next call; PythonComm.H:31-36 copies it out at once -- and no longer leaks one array per step).

MPI: with ``mpi4py`` importable and more than one rank the cell arrays are gathered to rank 0, solved
there and scattered back (python_module.py:179-191, 258-264, 501-511); otherwise everything runs on the
calling process (the serial module, singleCore/test_Case/python_module.py:139,199).
"""
import os
import sys
import traceback

import numpy as np

_home = os.environ.get("PSM_AMD_HOME") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _home not in sys.path:
    sys.path.insert(0, _home)

import psm_amd  # noqa: E402
from psm_amd import formats  # noqa: E402
from psm_amd.surrogate import SolverModule  # noqa: E402
from psm_amd.synthetic import SurrogateModel  # noqa: E402

comm, rank, nprocs = None, 0, 1
try:                                                        # python_module.py:13-16
    import mpi4py
    mpi4py.rc.initialize = False                            # OpenFOAM's Pstream owns MPI_Init / MPI_Finalize
    mpi4py.rc.finalize = False
    from mpi4py import MPI
    # The serial solvers (singleCore/DLPoissonSolver_*) never call MPI_Init: touching COMM_WORLD there would abort the
    # run on a host that merely has mpi4py installed.  Only an MPI that the solver has initialised is used.
    if MPI.Is_initialized():
        comm = MPI.COMM_WORLD
        rank, nprocs = comm.Get_rank(), comm.Get_size()
except ImportError:                                         # serial solver, no mpi4py
    pass


def load_case(directory: str = "."):
    """python_module.py:103-118,170: -> (SurrogateModel, maxs[4])."""
    pcainput = formats.load_pca(formats.find_pca(directory, "ipca_input_more"))
    pcap = formats.load_pca(formats.find_pca(directory, "ipca_p_more"))
    maxs = formats.read_maxs(os.path.join(directory, "maxs"))
    maxs_PCA = formats.read_maxs(os.path.join(directory, "maxs_PCA"))
    PC_p = formats.select_num_pc(pcap.explained_variance_ratio_, 0.95, None)            # :112
    PC_input = formats.select_num_pc(pcainput.explained_variance_ratio_, 0.995, None)   # :113
    weights = formats.read_keras_dense_weights(os.path.join(directory, "weights.h5"))   # :170
    if weights[0][0].shape[0] != PC_input or weights[-1][0].shape[1] != PC_p:
        raise ValueError(f"weights.h5 is {weights[0][0].shape[0]} -> {weights[-1][0].shape[1]}, "
                         f"the PCA files give {PC_input} -> {PC_p}")
    m = SurrogateModel("chapter5", 3, 1, pcainput.components_[:PC_input], pcainput.mean_, pcap.components_[:PC_p],
                       pcap.mean_, list(weights), scaler_kind="max_abs")
    m.in_a, m.out_a = float(maxs_PCA[0]), float(maxs_PCA[1])                             # :110, 351, 365
    return m, maxs[:4]


_module = None
len_rankwise = None
if rank == 0:                                               # python_module.py:168-170: only rank 0 holds the model
    _model, _maxs = load_case(os.getcwd())
    _module = SolverModule(_model, _maxs, device=int(os.environ.get("PSM_DEVICE", "0")))


def _gather(a):
    return [a] if comm is None or nprocs == 1 else comm.gather(a, root=0)


def init_func(array, top_boundary, obst_boundary, placeholder=0):
    """python_module.py:172-247 (serial: 3 arguments, singleCore python_module.py:139)."""
    global len_rankwise
    array_global, top_global, obst_global = _gather(np.asarray(array)), _gather(np.asarray(top_boundary)), _gather(np.asarray(obst_boundary))
    lens = _gather(np.asarray(array).shape[0])
    if rank == 0:
        len_rankwise = lens
        _pin_state.update(ptr=None, n=0, array=None, out=None)           # a new geometry = a new handle: nothing is registered on it yet
        _cat.update(buf=None, out=None)
        _module.init_func(np.concatenate(array_global), np.concatenate(top_global), np.concatenate(obst_global))
    return 0


_pin_state = {"ptr": None, "n": 0, "array": None, "out": None}
_PIN = os.environ.get("PSM_PIN_SOLVER_BUFFERS", "0") not in ("", "0")


def _solve_rank0(array):
    """rank 0: one step.  With PSM_PIN_SOLVER_BUFFERS=1 on the serial solver the caller's (persistent) array and one output
    array are registered on first use and re-registered if the caller ever comes with another buffer."""
    if not (_PIN and (comm is None or nprocs == 1) and array.flags.c_contiguous):
        return _module.py_func(array)
    ptr, n = array.ctypes.data, array.shape[0]
    # Same address and length, but another owner (array.base) than the one registered: the old buffer may have been freed and
    # this one allocated in its place -- register again.  (A solver that wraps ITS persistent C buffer in a fresh ndarray per
    # call has base None both times and keeps the registration, which is the case this option exists for.)
    prev = _pin_state["array"]
    stale = prev is not None and prev is not array and (getattr(prev, "base", None) is not getattr(array, "base", None))
    if _pin_state["ptr"] != ptr or _pin_state["n"] != n or stale:
        if _pin_state["ptr"] is not None:
            _module.unpin()
        out = np.empty(n, np.float64)
        _module.pin(array, out)
        _pin_state.update(ptr=ptr, n=n, array=array, out=out)          # `array` kept: the registration must not outlive the view
    return _module.py_func(array, out=_pin_state["out"])


_cat = {"buf": None, "out": None}


def _gathered(parts):
    """Parallel solver, rank 0: the ranks' arrays concatenated into ONE persistent buffer that this module owns and registers
    with the GPU (safe by construction: it lives as long as the module) -- the concatenation is the only host copy of the step."""
    n = sum(a.shape[0] for a in parts)
    if _cat["buf"] is None or _cat["buf"].shape[0] != n:
        if _cat["buf"] is not None:
            _module.unpin()
        _cat["buf"], _cat["out"] = np.empty((n, 5), np.float64), np.empty(n, np.float64)
        _module.pin(_cat["buf"], _cat["out"])
    np.concatenate(parts, out=_cat["buf"])
    return _cat["buf"]


def py_func(array_in, placeholder=0):
    """python_module.py:249-517: cells [N_local,5] -> p [N_local].  A failure on the GPU side never aborts the
    solver: it is reported and the previous pressure (column 4) is returned for this step."""
    array_in = np.asarray(array_in, np.float64)
    array_global = _gather(array_in)
    p_rankwise = None
    if rank == 0:
        try:
            if len(array_global) == 1:
                array = array_global[0]
                p = _solve_rank0(array)
            else:
                array = _gathered(array_global)
                p = _module.py_func(array, out=_cat["out"])
        except Exception:                                   # singleCore python_module.py:440-444 swallows and returns 0
            traceback.print_exc()
            array = np.asarray(array_global[0] if len(array_global) == 1 else np.concatenate(array_global), np.float64)
            p = array[:, 4].copy() if array.ndim == 2 and array.shape[1] >= 5 else np.zeros(array.shape[0] if array.ndim else 0)
        p_rankwise, init = [], 0
        own = len(array_global) > 1                         # slices of the module's persistent output buffer: handed out as copies
        for length in len_rankwise:                         # :501-507
            part = p[init:init + length, ...]
            p_rankwise.append(part.copy() if own else part)
            init += length
    if comm is None or nprocs == 1:
        return p_rankwise[0]
    return comm.scatter(p_rankwise, root=0)                 # :511
