"""compute_in_block_error (Improved_SM/deltaU_to_deltaP/source/pressureSM_deltas/utils.py:210-243, called at SM_call.py:555-557
and summarised as BIAS_block / RSME_block / STDE_block at SM_call.py:824-826): the error of the decoded blocks against the
de-meaned label blocks before any reassembly.  Golden values (tests/golden/block_error_deltas.npz) come from the reference's
own function run on what its timeStep statements produced (tests/golden/make_golden.py run_deltas(block_error=True))."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate
from test_gpu_parity import oracle_model

GOLD = cases.load_golden("block_error_deltas")


def _oracle_values(name):
    grid, model = cases.build(name)
    sp = cases.GOLDEN_CASES[name]
    om = oracle_model(model)
    sol = orc.solve_grid(grid, om, degenerate="strict")
    lay = orc.block_layout("deltas", grid.shape[0], grid.shape[1])
    yb = orc.label_blocks(grid[..., :3], grid[..., 3], lay, 3)
    scale = sp.get("max_abs_p", 1.0) * sp.get("U_max_norm", 1.0) ** 2
    return orc.compute_in_block_error(sol.block_pred, yb * scale, sol.x_blocks[..., 2:3] != 0)


@pytest.mark.parametrize("name", cases.BLOCK_ERROR_CASES)
def test_oracle_block_error_matches_the_reference_function(name):
    a, b = _oracle_values(name)
    # float64 on both sides; the oracle's network output differs from the stand-in's in the last float32 bits
    assert abs(a - GOLD[name][0]) <= 1e-6 * np.sqrt(GOLD[name][1]) and abs(b - GOLD[name][1]) <= 1e-6 * GOLD[name][1]


def test_oracle_block_error_nan_rules():
    """NaN differences are left out of the means (utils.py:224); a NaN label makes the norm -- np.max -- NaN."""
    rng = np.random.default_rng(0)
    t, p = rng.standard_normal((2, 4, 4, 1)), rng.standard_normal((2, 4, 4, 1))
    fb = rng.random((2, 4, 4, 1)) > 0.3
    a, b = orc.compute_in_block_error(p, t, fb)
    p2 = p.copy()
    p2[fb][0] = np.nan                                            # (a copy: no effect) -- now a real NaN on a flow cell
    idx = tuple(np.argwhere(fb)[0])
    p2[idx] = np.nan
    a2, b2 = orc.compute_in_block_error(p2, t, fb)
    keep = fb.copy(); keep[idx] = False
    norm = t[fb].max() - t[fb].min()
    assert abs(a2 - np.mean((p - t)[keep]) / norm) < 1e-12 and np.isfinite(b2) and (a2 != a)
    t2 = t.copy(); t2[idx] = np.nan
    assert np.isnan(orc.compute_in_block_error(p, t2, fb)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", cases.BLOCK_ERROR_CASES)
@pytest.mark.parametrize("bound", [False, True])
def test_gpu_block_error_golden(name, bound):
    """GridSurrogate.block_error after a solve -- on the general path (decoded blocks stored) and on a bound geometry (blocks
    decoded again from the stored network output) -- against the reference function's values."""
    grid, model = cases.build(name)
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, grid.shape[0], grid.shape[1]) as sur:
        if bound:
            assert sur.bind_geometry(g32[..., :3])
        sur.solve(g32[..., :3], out_scale=[model.out_scale])
        m = sur.block_error(g32[..., :3], g32[..., 3])
        assert sur.geometry_bound == bound
    ga, gb = GOLD[name]
    assert abs(m["mean_err"] - ga) <= 2e-4 * np.sqrt(gb) and abs(m["mean_sq_err"] - gb) <= 2e-4 * gb
    assert abs(m["rmseNorm"] - np.sqrt(gb) * 100) <= 1e-2 and m["n"] > 0 and m["normVal"] > 0


@pytest.mark.gpu
def test_gpu_block_error_needs_a_solve_and_skips_nan_differences():
    grid, model = cases.build("deltas_256x256")
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        with pytest.raises(Exception):
            sur.block_error(g32[..., :3], g32[..., 3])
        sur.solve(g32[..., :3], out_scale=[model.out_scale])
        m0 = sur.block_error(g32[..., :3], g32[..., 3])
        lab = g32[..., 3].copy()
        lab[100, 100] = np.nan                                    # a NaN label on a flow cell: norm = np.max(...) = NaN
        assert g32[100, 100, 2] != 0
        m1 = sur.block_error(g32[..., :3], lab)
        assert np.isnan(m1["mean_err"]) and np.isnan(m1["normVal"]) and m1["n"] < m0["n"]


@pytest.mark.gpu
def test_gpu_block_error_refuses_a_solve_that_ran_on_the_ring():
    """the asynchronous ring solves on its slots' workspaces, so the handle's own workspace would hold an OLDER
    network output -- PSM_ERR_STATE instead of metrics of stale data; a synchronous solve afterwards makes it valid again."""
    grid, model = cases.build("deltas_256x256")
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        sur.solve(g32[..., :3], out_scale=[model.out_scale])
        m0 = sur.block_error(g32[..., :3], g32[..., 3])
        other = (g32[..., :3] * np.float32(0.5)).copy(); other[..., 2] = g32[..., 2]
        sur.wait(sur.submit(other, out_scale=[model.out_scale]))
        with pytest.raises(Exception, match="ring"):
            sur.block_error(g32[..., :3], g32[..., 3])
        sur.solve(g32[..., :3], out_scale=[model.out_scale])
        m1 = sur.block_error(g32[..., :3], g32[..., 3])
        assert m0["mean_sq_err"] == m1["mean_sq_err"]
