"""The drop-in `python_module` (init_func / py_func looked up by the unmodified solver,
PythonComm_init.H:11-18): artefact loading from the working directory with the reference's file names and
PC-count rule (python_module.py:103-118,170), and -- on the GPU -- the same pressures as SolverModule."""
import importlib
import os
import pickle
import sys

import numpy as np
import pytest

import cases
import h5write
from psm_amd import SolverModule


@pytest.fixture()
def case_dir(tmp_path, monkeypatch):
    from sklearn.decomposition import PCA
    array, top, obst, model, maxs = cases.build_mesh_case()
    P = 40
    full = cases.synthetic.make_model("chapter5", p_in=P, p_out=P, seed_pca=4321, seed_w=11)
    model.comp_in, model.comp_out = full.comp_in[:32], full.comp_out[:32]
    model.mean_in, model.mean_out = full.mean_in, full.mean_out
    # stored objects carry 40 components; the bare argmax rule keeps 32 of each (python_module.py:112-113)
    evr_in = np.array([0.9949 / 32] * 32 + [0.0051 / 8] * 8)
    evr_p = np.array([0.9499 / 32] * 32 + [0.05 / 8] * 8)
    for stem, comp, mean, evr in (("ipca_input_more", full.comp_in, full.mean_in, evr_in), ("ipca_p_more", full.comp_out, full.mean_out, evr_p)):
        o = PCA(n_components=P)
        o.components_, o.mean_, o.explained_variance_ratio_ = comp, mean, evr
        with open(tmp_path / (stem + ".pkl"), "wb") as f:
            pickle.dump(o, f)
    np.savetxt(tmp_path / "maxs", np.array(maxs))
    np.savetxt(tmp_path / "maxs_PCA", np.array([model.in_a, model.out_a]))
    h5write.write_keras_dense(str(tmp_path / "weights.h5"), model.weights)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("PSM_AMD_HOME", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return array, top, obst, model, maxs


def _import_module():
    sys.modules.pop("psm_amd.python_module", None)
    return importlib.import_module("psm_amd.python_module")


def test_module_loads_the_case_like_the_reference(case_dir):
    array, top, obst, model, maxs = case_dir
    pm = _import_module()
    assert callable(pm.init_func) and callable(pm.py_func) and pm.rank == 0
    m = pm._module.model
    assert (m.p_in, m.p_out) == (32, 32)
    np.testing.assert_array_equal(m.comp_in, model.comp_in)         # first 32 of the 40 stored components
    np.testing.assert_array_equal(m.comp_out, model.comp_out)
    assert (m.in_a, m.out_a) == (model.in_a, model.out_a)
    np.testing.assert_allclose(pm._module.maxs, maxs)
    os.remove("maxs_PCA")
    with pytest.raises(OSError):
        _import_module()


@pytest.mark.gpu
def test_module_py_func_equals_solver_module(case_dir):
    array, top, obst, model, maxs = case_dir
    pm = _import_module()
    assert pm.init_func(array, top, obst) == 0                      # serial solver: three arguments
    p = pm.py_func(array)
    ref = SolverModule(model, maxs)
    ref.init_func(array, top, obst)
    np.testing.assert_array_equal(p, ref.py_func(array))
    assert pm.init_func(array, top, obst, 0) == 0                   # parallel solver: rank argument
    np.testing.assert_array_equal(pm.py_func(array, 0), p)
    bad = array.copy(); bad = bad[:-5]                              # wrong cell count: reported, previous p returned
    pm.len_rankwise = [bad.shape[0]]
    np.testing.assert_array_equal(pm.py_func(bad, 0), bad[:, 4])


@pytest.mark.gpu
def test_module_pins_the_solvers_persistent_buffer_when_asked(case_dir, monkeypatch):
    """PSM_PIN_SOLVER_BUFFERS=1 (serial solver): the array the solver hands over every step is registered once, later steps
    with the same buffer take the copy-free path and return the same persistent output array; another buffer re-registers;
    the pressures are those of the default path bit for bit."""
    array, top, obst, model, maxs = case_dir
    ref_mod = _import_module()
    ref_mod.init_func(array, top, obst)
    ref = [ref_mod.py_func(array * s).copy() for s in (1.0, 1.01)]
    monkeypatch.setenv("PSM_PIN_SOLVER_BUFFERS", "1")
    pm = _import_module()
    assert pm._PIN and pm.init_func(array, top, obst) == 0
    buf = np.ascontiguousarray(array, np.float64).copy()              # the solver's input_vals: one allocation for the whole run
    p0 = pm.py_func(buf)
    assert pm._pin_state["ptr"] == buf.ctypes.data and pm._module._pinned is not None
    np.testing.assert_array_equal(p0, ref[0])
    buf[:] = array * 1.01                                             # next time step, written in place
    p1 = pm.py_func(buf)
    assert np.shares_memory(p1, p0) and pm._pin_state["ptr"] == buf.ctypes.data       # the same registered output array
    np.testing.assert_array_equal(p1, ref[1])
    other = np.ascontiguousarray(array, np.float64).copy()            # a different buffer: registered in its place
    np.testing.assert_array_equal(pm.py_func(other), ref[0])
    assert pm._pin_state["ptr"] == other.ctypes.data
    pm.init_func(array, top, obst)                                    # a new geometry forgets the registration
    assert pm._pin_state["ptr"] is None
    np.testing.assert_array_equal(pm.py_func(other), ref[0])
