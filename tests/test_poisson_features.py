"""pressureSM_Poisson input features (pressureSM_Poisson/SM_call.py:588-711): oracle vs the reference's own
statements (golden fixture written by tests/golden/make_golden.py), the HIP kernels vs both, and the
four-channel deltas surrogate (mask = channel 3) fed with them."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import EvaluationPoisson, GridSurrogate, synthetic
from test_oracle_golden import oracle_model


def _oracle_grid(c):
    return orc.poisson_features(c["ux"], c["uy"], c["dux"], c["duy"], c["sdfunct"], c["L"], c["U"], c["k"], c["max_abs"])


def _model4():
    m = synthetic.make_model("deltas", p_in=48, p_out=40, c_in=4, seed_pca=777, seed_w=5)
    m.sdf_ch = 3
    return m


def test_oracle_features_match_reference_run():
    c, gold = cases.build_poisson_case(), cases.load_golden("poisson_features_160x200")
    grid, term = _oracle_grid(c)
    np.testing.assert_allclose(term[48:112, 40:104], gold["term_crop"], rtol=1e-14, atol=0)
    assert np.isclose(term.sum(), float(gold["term_sum"]), rtol=1e-12)
    assert np.isclose(np.abs(term).sum(), float(gold["term_abs_sum"]), rtol=1e-12)
    assert np.abs(grid - gold["grid"]).max() <= 2e-7 * np.abs(gold["grid"]).max()    # fixture stored as float32
    # structure: the source term vanishes on solids and next to them, the SDF channel is the image itself
    solid = c["sdfunct"] == 0
    assert np.all(term[solid] == 0) and np.all(grid[..., 3] == c["sdfunct"] / c["max_abs"][3])


def test_masked_gradient_is_np_gradient_away_from_solids():
    rng = np.random.default_rng(3)
    f = rng.standard_normal((40, 50))
    gy, gx = orc.masked_gradient(f)
    ry, rx = np.gradient(f)
    np.testing.assert_array_equal(gy, ry); np.testing.assert_array_equal(gx, rx)
    f[10, 20] = np.nan
    gy, gx = orc.masked_gradient(f)
    for (i, j) in ((10, 20), (9, 20), (11, 20), (10, 19), (10, 21)):
        assert gy[i, j] == 0 and gx[i, j] == 0
    assert gy[9, 19] == ry[9, 19] and gx[12, 20] == rx[12, 20]     # diagonal / second neighbours untouched


@pytest.mark.gpu
def test_gpu_features_match_reference_run_and_oracle():
    c, gold = cases.build_poisson_case(), cases.load_golden("poisson_features_160x200")
    ref, _ = _oracle_grid(c)
    with GridSurrogate(_model4(), 160, 200) as sur:
        g = sur.poisson_features(c["ux"], c["uy"], c["dux"], c["duy"], c["sdfunct"], c["L"], c["U"], c["k"], c["max_abs"])
        assert g.shape == (160, 200, 4) and g.dtype == np.float32
        # float64 arithmetic on the device, float32 image: one float32 rounding of the reference's values
        assert np.abs(g - gold["grid"]).max() <= 3e-7 * np.abs(gold["grid"]).max()
        assert np.abs(g - ref).max() <= 3e-7 * np.abs(ref).max()
        with pytest.raises(Exception):
            sur.poisson_features(c["ux"], c["uy"], c["dux"], c["duy"], c["sdfunct"], c["L"], 0.0, c["k"], c["max_abs"])


@pytest.mark.gpu
def test_gpu_poisson_evaluator_end_to_end():
    """EvaluationPoisson.timeStep_grid = features -> 4-channel deltas surrogate (mask channel 3) -> assembly
    (-> deltaU-change weighting), against the oracle chain on the same inputs."""
    c = cases.build_poisson_case()
    model = _model4()
    ev = EvaluationPoisson(5e-3, 128, 32, 0.95, 0.95, None, None, 128, "std", c["k"], None, model=model,
                           max_abs=tuple(c["max_abs"]) + (0.51,))
    field = ev.timeStep_grid(c["ux"], c["uy"], c["dux"], c["duy"], c["sdfunct"], c["L"], c["U"], apply_deltaU_change_wgt=False)
    grid, _ = _oracle_grid(c)
    om = oracle_model(model)
    om.out_scale = 0.51 * c["U"] ** 2
    sol = orc.solve_grid(grid, om)
    assert sol.fields.shape == (160, 200, 1)
    assert np.abs(field - sol.fields[..., 0]).max() <= 1e-4 * np.abs(sol.fields).max()
    # with the weighting (SM_call.py:843-848): field = previous + gaussian(change * gaussian(dU change))
    import scipy.ndimage as ndi
    rng = np.random.default_rng(9)
    dU = np.abs(rng.standard_normal((160, 200))); dU /= dU.max()
    prev = 0.1 * rng.standard_normal((160, 200))
    got = ev.timeStep_grid(c["ux"], c["uy"], c["dux"], c["duy"], c["sdfunct"], c["L"], c["U"], dU, prev)
    w = ndi.gaussian_filter(dU, sigma=(50, 50), order=0)
    want = prev + ndi.gaussian_filter((sol.fields[..., 0] - prev) * w, sigma=(10, 10), order=0)
    assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), np.abs(sol.fields).max())
