"""oracle/psm_cpu.c -- the C / OpenMP restatement of the surrogate path ("CPU back-end A" of BASELINE.md, the CPU baseline
bench.py times): pinned like the NumPy oracle, by every golden case produced from the reference's own statements, and
checked against the NumPy oracle on the BASELINE configs."""
import numpy as np
import pytest

import cases
from oracle import psm_cpu, psm_oracle as orc
from psm_amd import synthetic
from test_oracle_golden import oracle_model


@pytest.mark.parametrize("name", cases.DENSE_GOLDEN_CASES)
def test_c_port_matches_reference_run(name):
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    f, x = psm_cpu.solve_grid(grid, psm_cpu.CpuModel(oracle_model(model), strict=True), want_x_input=True)
    assert x.shape[0] == int(gold["n_blocks"])
    np.testing.assert_allclose(x, gold["x_input"], rtol=1e-9, atol=1e-11)          # float64 PCA encode + scaler
    ref = gold["fields"]
    np.testing.assert_allclose(f, ref, rtol=1e-6, atol=2e-6 * np.abs(ref).max())    # float32 network in between


@pytest.mark.parametrize("variant,ny,nx", [("gradp", 256, 256), ("deltas", 256, 256), ("deltas", 512, 512), ("chapter5", 128, 128)])
def test_c_port_matches_numpy_oracle_on_baseline_shapes(variant, ny, nx):
    """BASELINE configs 1, 2, 4 (shape) and 0: including the grids where p_i == 0 (build-defined skip mode)."""
    model = synthetic.make_model(variant, p_in=48, p_out=40)
    g = synthetic.channel_grid(ny, nx, seed=5) if variant != "chapter5" else synthetic.cavity_grid(ny)
    om = oracle_model(model)
    sol = orc.solve_grid(g, om)
    cm = psm_cpu.CpuModel(om)
    f1 = psm_cpu.solve_grid(g, cm, threads=1)
    assert np.abs(f1 - sol.fields).max() <= 2e-6 * np.abs(sol.fields).max()
    f4 = psm_cpu.solve_grid(g, cm, threads=4)
    assert np.abs(f4 - f1).max() <= 1e-9 * np.abs(f1).max()                         # only the split-K summation order differs


def test_c_port_strict_mode_on_the_degenerate_golden():
    grid, model = cases.build("gradp_degenerate_256x256")
    ref = cases.load_golden("gradp_degenerate_256x256")["fields"]
    f = psm_cpu.solve_grid(grid, psm_cpu.CpuModel(oracle_model(model), strict=True))
    np.testing.assert_array_equal(np.isnan(f), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(f[ok], ref[ok], rtol=1e-6, atol=2e-6 * np.abs(ref[ok]).max())
    grid, model = cases.build("deltas_degenerate_512x512")
    with pytest.raises(ValueError):
        psm_cpu.solve_grid(grid, psm_cpu.CpuModel(oracle_model(model), strict=True))


def test_c_port_refuses_what_the_reference_cannot_process():
    model = synthetic.make_model("deltas", p_in=8, p_out=8)
    g = synthetic.channel_grid(512, 512, seed=1)                                    # p_i == 0: broadcast error at SMD:335
    with pytest.raises(ValueError):
        psm_cpu.solve_grid(g, psm_cpu.CpuModel(oracle_model(model), strict=True))
    with pytest.raises(ValueError):
        psm_cpu.solve_grid(g[:100], psm_cpu.CpuModel(oracle_model(model)))           # smaller than one block
