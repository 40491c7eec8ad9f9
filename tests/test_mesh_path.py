"""Solver boundary (cells[N,5] float64 -> p[N] float64; PythonComm.H:2-9,31-36):
oracle vs the outputs of the reference's whole rank-0 ``py_func`` body, host tables of the
product vs the oracle, and (GPU) the C-ABI ``psm_set_geometry`` / ``psm_solve`` path."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import _lib, SolverModule, geometry
from test_oracle_golden import oracle_model


@pytest.fixture(scope="module")
def mesh_case():
    array, top, obst, model, maxs = cases.build_mesh_case()
    geo = orc.init_geometry(array, top, obst)
    return array, top, obst, model, maxs, geo, cases.load_golden("mesh_chapter5")


def test_init_tables_match_the_reference_helpers(mesh_case):
    array, top, obst, model, maxs, geo, gold = mesh_case
    # create_uniform_grid / interp_weights of the reference file, executed by make_golden.py
    assert geo.ny * geo.nx == int(gold["ref_grid_n"])
    X0, Y0 = orc.create_uniform_grid(round(array[:, 2].min(), 2), round(array[:, 2].max(), 2),
                                     round(array[:, 3].min(), 2), round(array[:, 3].max(), 2), 5e-3)
    assert np.isclose(X0.sum(), float(gold["ref_X0_sum"]), rtol=1e-13)
    assert np.isclose(Y0.sum(), float(gold["ref_Y0_sum"]), rtol=1e-13, atol=1e-9)
    assert int(geo.vert_m2g.astype(np.int64).sum()) == int(gold["ref_v1_sum"])
    assert int(geo.vert_g2m.astype(np.int64).sum()) == int(gold["ref_v2_sum"])
    assert np.isclose(np.abs(geo.wts_m2g).sum(), float(gold["ref_w1_abs_sum"]), rtol=1e-12)
    assert np.isclose(np.abs(geo.wts_g2m).sum(), float(gold["ref_w2_abs_sum"]), rtol=1e-12)


def test_oracle_py_func_matches_reference_run(mesh_case):
    array, top, obst, model, maxs, geo, gold = mesh_case
    p, grid, sol = orc.py_func_mesh(array, geo, oracle_model(model), maxs)
    assert sol.x_blocks.shape[0] == int(gold["n_blocks"])
    np.testing.assert_allclose(grid, gold["grid"], rtol=0, atol=2e-7)          # fixture stored as float32
    np.testing.assert_allclose(p, gold["p"], rtol=1e-6, atol=1e-6 * np.abs(gold["p"]).max())
    fb = p == array[:, 4]
    assert 0 < fb.sum() < len(p)                                               # near-wall / NaN fallback exercised


def test_product_host_tables_equal_oracle(mesh_case):
    array, top, obst, model, maxs, geo, gold = mesh_case
    t = geometry.build_geometry(array, top, obst)
    assert (t.ny, t.nx) == (geo.ny, geo.nx)
    np.testing.assert_array_equal(t.vtx_m2g, geo.vert_m2g)
    np.testing.assert_array_equal(t.vtx_g2m, geo.vert_g2m)
    np.testing.assert_allclose(t.wts_m2g, geo.wts_m2g, rtol=0, atol=1e-14)
    np.testing.assert_allclose(t.wts_g2m, geo.wts_g2m, rtol=0, atol=1e-14)
    np.testing.assert_array_equal(t.indices, geo.indices)
    np.testing.assert_allclose(t.sdfunct, geo.sdfunct, rtol=0, atol=1e-14)


@pytest.mark.gpu
def test_solver_module_init_func_py_func(mesh_case):
    array, top, obst, model, maxs, geo, gold = mesh_case
    sm = SolverModule(model, maxs)
    assert sm.init_func(array, top, obst, 0) == 0
    p = sm.py_func(array, 0)
    assert p.dtype == np.float64 and p.shape == (len(array),)
    ref = gold["p"]
    # cells that keep the previous pressure are bit-identical; the others within the f32 tolerance
    fb = ref == array[:, 4]
    np.testing.assert_array_equal(p[fb], array[fb, 4])
    assert np.abs(p - ref).max() <= 2e-4 * np.abs(ref).max()
    # a second time step through the same handle (different fields, same geometry)
    array2, _, _ = cases.synthetic.channel_mesh(step=3)
    p2 = sm.py_func(array2, 0)
    ref2, _, _ = orc.py_func_mesh(array2, geo, oracle_model(model), maxs)
    assert np.abs(p2 - ref2).max() <= 2e-4 * np.abs(ref2).max()
    with pytest.raises(Exception):
        sm.py_func(array2[:-5], 0)              # wrong cell count: reported, not fatal


@pytest.mark.gpu
def test_init_func_binds_the_geometry_for_py_func_only(mesh_case, monkeypatch):
    """psm_set_geometry binds the obstacle for the mesh entry (6-launch solves); the pressures equal those of the
    general path (PSM_NO_BIND=1) to float32 summation order, and grid-native solves on the same handle stay general."""
    array, top, obst, model, maxs, geo, gold = mesh_case
    sm = SolverModule(model, maxs)
    sm.init_func(array, top, obst, 0)
    assert sm._sur.geometry_bound
    p_bound = sm.py_func(array, 0)
    g = cases.synthetic.channel_grid(sm.tables.ny, sm.tables.nx, seed=9).astype(np.float32)    # another geometry
    f_grid = sm._sur.solve(g)[0]
    monkeypatch.setenv("PSM_NO_BIND", "1")
    sm2 = SolverModule(model, maxs)
    sm2.init_func(array, top, obst, 0)
    assert not sm2._sur.geometry_bound
    p_general = sm2.py_func(array, 0)
    assert np.abs(p_bound - p_general).max() <= 2e-5 * np.abs(p_general).max()
    np.testing.assert_allclose(f_grid, sm2._sur.solve(g)[0], rtol=0, atol=1e-6 * np.abs(f_grid).max())


@pytest.mark.gpu
def test_pinned_solver_buffers_give_the_same_pressures(mesh_case):
    """psm_pin_buffers: the solver's persistent arrays registered for direct DMA -- bit-identical results, staging path
    still taken for any other pointer, unpin restores it."""
    array, top, obst, model, maxs, geo, gold = mesh_case
    sm = SolverModule(model, maxs)
    sm.init_func(array, top, obst)
    ref = sm.py_func(array)
    cells, out = np.ascontiguousarray(array, np.float64).copy(), np.empty(array.shape[0], np.float64)
    sm.pin(cells, out)
    for step in range(3):
        cells[:, 0] = array[:, 0] * (1.0 + 0.01 * step)              # the solver overwrites its buffer in place
        got = sm.py_func(cells, out=out)
        assert got is out
        np.testing.assert_array_equal(got, sm.py_func(cells.copy()))   # other pointer -> staging path, same numbers
    cells[:, 0] = array[:, 0]
    np.testing.assert_array_equal(sm.py_func(cells, out=out), ref)
    sm.unpin()
    np.testing.assert_array_equal(sm.py_func(cells, out=out), ref)


@pytest.mark.gpu
def test_ensemble_of_cases_advanced_from_one_thread(mesh_case):
    """psm_solve_begin / psm_solve_end: three independent cases (one handle = one geometry and stream each) advanced
    in lock-step from one thread -- begin on all, end on all -- give the pressures of the synchronous calls; a second
    begin without an end, and an end without a begin, are refused."""
    array, top, obst, model, maxs, geo, gold = mesh_case
    mods, arrays = [], []
    for k in range(3):
        a = cases.build_mesh_case(step=k)[0]
        sm = SolverModule(model, maxs)
        sm.init_func(a, top, obst)
        mods.append(sm); arrays.append(a)
    ref = [sm.py_func(a) for sm, a in zip(mods, arrays)]
    for rep in range(3):
        for sm, a in zip(mods, arrays):
            sm.py_func_begin(a)
        got = [sm.py_func_end() for sm in mods]
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g, r)
    mods[0].py_func_begin(arrays[0])
    with pytest.raises(_lib.PsmError):
        mods[0].py_func_begin(arrays[0])
    mods[0].py_func_end()
    with pytest.raises(_lib.PsmError):
        mods[0].py_func_end()


def test_oracle_filters_match_reference_run():
    """assemble_prediction with apply_filter / apply_deltaU_change_wgt (SM_call.py:352-363): the oracle's
    reassembly followed by SciPy's gaussian_filter (the routine the reference calls) against the run of
    the reference's own method."""
    import scipy.ndimage as ndi
    grid, model, bp, dU, dPprev = cases.build_filter_case()
    gold = cases.load_golden("deltas_filters_256x256")
    lay = orc.block_layout("deltas", *grid.shape[:2])
    a = orc.assemble_deltas(bp, orc.extract_blocks(grid, lay, 3), lay)
    res = ndi.gaussian_filter(a.field, sigma=(10, 10), order=0)
    w = ndi.gaussian_filter(dU, sigma=(50, 50), order=0)
    chg = ndi.gaussian_filter((res - dPprev) * w, sigma=(10, 10), order=0)
    np.testing.assert_allclose(res, gold["result"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(chg, gold["change"], rtol=0, atol=2e-7)


@pytest.mark.gpu
def test_gpu_filters_and_weighting():
    from psm_amd import Evaluation
    grid, model, bp, dU, dPprev = cases.build_filter_case()
    gold = cases.load_golden("deltas_filters_256x256")
    lay = orc.block_layout("deltas", *grid.shape[:2])
    ev = Evaluation(5e-3, 128, 32, 0.95, 0.95, None, None, 128, "std", model=model)
    ev.x_array = orc.extract_blocks(grid, lay, 3)
    res, chg = ev.assemble_prediction(bp, [list(t) for t in lay.tags], lay.n_x, lay.n_y, True, 256, 256, dU, dPprev, True)
    assert np.abs(res - gold["result"]).max() <= 1e-4 * np.abs(gold["result"]).max()
    assert np.abs(chg - gold["change"]).max() <= 1e-4 * max(np.abs(gold["change"]).max(), 1e-3)
    # the filter alone against SciPy, including a radius larger than the image (sigma 50 -> radius 200 on 131 rows)
    import scipy.ndimage as ndi
    rng = np.random.default_rng(3)
    f = rng.standard_normal((131, 257)).astype(np.float32)
    for sig in ((10, 10), (50, 50), (2.5, 7.0)):
        got = ev._surrogate(256, 256).gaussian_filter(f, sig)
        ref = ndi.gaussian_filter(f.astype(np.float64), sigma=sig, order=0)
        assert np.abs(got - ref).max() <= 2e-6


def _integ_setup():
    ic = cases.build_integration_case()
    Ny, Nx = ic["sdfunct"].shape
    cx, cy = orc.integration_center(ic["sdfunct"], ic["min_x"], ic["max_x"], ic["X0"].min(), ic["delta"])
    dx = (ic["max_x"] - ic["min_x"]) / (Nx - 1)          # np.diff(xl)[0] of UGP:594
    dy = (ic["max_y"] - ic["min_y"]) / (Ny - 1)
    return ic, cx, cy, dx, dy


def test_oracle_integration_matches_reference_run():
    ic, cx, cy, dx, dy = _integ_setup()
    gold = cases.load_golden("gradp_integration_320x384")
    assert (cx, cy) == (int(gold["center_p_x"]), int(gold["center_p_y"]))
    p = orc.integrate_gradp(ic["gradP"], ic["sdfunct"], dx, dy, cy, cx)
    np.testing.assert_allclose(p, gold["p"], rtol=0, atol=5e-7 * np.abs(gold["p"]).max() + 1e-7)


@pytest.mark.gpu
def test_gpu_integration_of_gradp():
    from psm_amd import GridSurrogate
    ic, cx, cy, dx, dy = _integ_setup()
    gold = cases.load_golden("gradp_integration_320x384")
    model = cases.synthetic.make_model("gradp", p_in=8, p_out=8)
    with GridSurrogate(model, 320, 384) as sur:
        sur.set_integration(ic["sdfunct"], cy, cx, dx, dy)
        p = sur.integrate_gradp(ic["gradP"])
        assert np.abs(p - gold["p"]).max() <= 1e-4 * np.abs(gold["p"]).max()
        # sdf values >= 1 exercise the reference's index quirk (int(sdf) = 1 -> a second fix-up index)
        sd2 = ic["sdfunct"].copy()
        sd2[10:40, 50:90] = 1.2
        sur.set_integration(sd2, cy, cx, dx, dy)
        p2 = sur.integrate_gradp(ic["gradP"])
        ref2 = orc.integrate_gradp(ic["gradP"], sd2, dx, dy, cy, cx)
        assert np.abs(p2 - ref2).max() <= 1e-4 * np.abs(ref2).max()


@pytest.mark.gpu
def test_solve_after_the_plan_was_dropped_is_a_state_error(mesh_case):
    """psm_set_geometry's tables belong to one plan.  A later psm_plan_grid with another size (or a psm_set_*
    call that drops the plan) must make psm_solve fail with PSM_ERR_STATE instead of launching on freed buffers; calling
    init_func again restores the path."""
    import ctypes as C
    array, top, obst, model, maxs, geo, gold = mesh_case
    sm = SolverModule(model, maxs)
    sm.init_func(array, top, obst, 0)
    ref = sm.py_func(array, 0)
    sur = sm._sur
    assert sur.lib.psm_plan_grid(sur.h, sm.tables.ny + 96, sm.tables.nx) == 0           # another grid: stale mesh tables
    cells = np.ascontiguousarray(array, np.float64)
    out = np.empty(len(array), np.float64)
    rc = sur.lib.psm_solve(sur.h, cells.ctypes.data_as(C.POINTER(C.c_double)), len(array), 0, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == -2
    grid = np.empty((sm.tables.ny, sm.tables.nx), np.float64)
    rc = sur.lib.psm_mesh_to_grid(sur.h, cells[:, 4].copy().ctypes.data_as(C.POINTER(C.c_double)), len(array), 1, 0, grid.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == -2
    sm2 = SolverModule(model, maxs)                     # a fresh handle is unaffected, and gives the same pressures
    sm2.init_func(array, top, obst, 0)
    np.testing.assert_array_equal(sm2.py_func(array, 0), ref)
