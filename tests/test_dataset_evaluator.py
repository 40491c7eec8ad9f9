"""`Evaluation(delta, shape, overlap, var_p, var_in, dataset_path, model_path, max_num_PC, standardization_method)`
driven from files in the reference's formats: padded HDF5 dataset, `maxs`, pickled PCA
objects, `mean_std.npz`, Keras-style Dense `.h5` -- `computeOnlyOnce(sim)` + `timeStep(sim, time, ...)`
(pressureSM_deltas/SM_call.py:89-180, 367-575) against the oracle chain on the same frame."""
import os

import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import Evaluation, formats, geometry, surrogate
from test_oracle_golden import oracle_model


@pytest.fixture(scope="module")
def ds(tmp_path_factory):
    d = tmp_path_factory.mktemp("artefacts")
    return str(d), cases.build_dataset_case(str(d))


def test_dataset_reader_and_padding(ds):
    d, c = ds
    with formats.PaddedDataset(c["dataset_path"]) as f:
        assert f.shape == c["sim"].shape
        data, top, obst = f.read(0, 1)
    assert data.shape == (1, 1) + c["sim"].shape[2:] and data.dtype == np.float32
    np.testing.assert_array_equal(data[0, 0], c["sim"][0, 1])
    assert formats.first_index(data[0, 0, :, 0], formats.PAD_VALUE) == c["N"]
    assert formats.first_index(top[0, 0, :, 0], formats.PAD_VALUE) == len(c["top"])
    assert formats.first_index(np.arange(5.0), -100.0) == 5            # no padding: build-defined (reference: TypeError)
    with pytest.raises(IndexError):
        formats.read_dataset(c["dataset_path"], 1, 0)
    with pytest.raises(formats.H5FormatError):
        formats.PaddedDataset(c["model_path"])                          # a valid HDF5 file without `sim_data`


def test_artifact_loader_follows_the_reference_rules(ds):
    d, c = ds
    m, maxs = surrogate.load_artifacts("deltas", c["model_path"], d, 0.95, 0.95, 128, "std", 128, 32)
    assert (m.p_in, m.p_out) == (24, 24)                                # argmax(cumsum > var) = 24, within (1, max_num_PC]
    np.testing.assert_array_equal(m.comp_in, c["model"].comp_in)
    np.testing.assert_array_equal(m.mean_out, c["model"].mean_out)
    np.testing.assert_allclose(maxs, cases.DATASET_MAXS)
    np.testing.assert_array_equal(m.in_b, c["model"].in_b)
    for (W, b), (W2, b2) in zip(m.weights, c["model"].weights):
        np.testing.assert_array_equal(W, W2); np.testing.assert_array_equal(b, b2)
    with pytest.raises(ValueError, match="network is 24 -> 24"):         # max_num_PC below the rule's count -> max_num_PC
        surrogate.load_artifacts("deltas", c["model_path"], d, 0.95, 0.95, 16, "std", 128, 32)
    # the dependency-free export carries the same three arrays
    formats.save_pca_npz(os.path.join(d, "copy.npz"), formats.load_pca(os.path.join(d, "ipca_p.pkl")))
    a, b = formats.load_pca(os.path.join(d, "copy.npz")), formats.load_pca(os.path.join(d, "ipca_p.pkl"))
    np.testing.assert_array_equal(a.components_, b.components_)
    np.testing.assert_array_equal(a.explained_variance_ratio_, b.explained_variance_ratio_)


def _tables(c):
    cells = np.asarray(c["sim"][0, 0, :c["N"]], np.float64)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)          # the dataset stores float32
    return geometry.build_geometry_evaluator(cells[:, 3:5], cells[:, 2], f32(c["top"]), f32(c["obst"]), 5e-3, idw_fallback=True)


def test_evaluator_geometry_rules(ds):
    """computeOnlyOnce's own rules (SM_call.py:103-169) on the host tables: 3-digit bounds, every 5th boundary
    point, `p` decides interpolability, zero-initialised indices."""
    d, c = ds
    t = _tables(c)
    assert (t.ny, t.nx) == (138, 300)          # bounds rounded to 3 digits (python_module.py rounds to 2: 140)
    assert t.vtx_g2m is None
    inside = t.indices.any(axis=1)
    assert np.all(t.sdfunct[t.indices[inside, 0], t.indices[inside, 1]] >= 0)
    assert (t.sdfunct > 0).sum() > 0.8 * t.ny * t.nx and (t.sdfunct == 0).sum() > 100      # obstacle + rim


def test_oracle_front_end_matches_reference_run(ds):
    """oracle.evaluator_grid_deltas against the reference's own statements (SM_call.py:381-451 with
    utils.interpolate_fill), executed by make_golden.py on frame 1 of this dataset: bit for bit."""
    d, c = ds
    gold = cases.load_golden("evaluator_grid_138x300")
    t = _tables(c)
    g, dU, dP, U = orc.evaluator_grid_deltas(c["sim"][0, 1, :c["N"]], t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct, cases.DATASET_MAXS)
    assert U == float(gold["U_max_norm"])
    np.testing.assert_array_equal(g[40:100, 120:200], gold["grid_crop"])
    np.testing.assert_allclose(g.sum(axis=(0, 1)), gold["grid_sum"], rtol=1e-13)
    np.testing.assert_allclose(np.abs(g).sum(axis=(0, 1)), gold["grid_abs_sum"], rtol=1e-13)
    np.testing.assert_array_equal(dU[40:100, 120:200], gold["dU_crop"])
    np.testing.assert_array_equal(dP[40:100, 120:200], gold["dPprev_crop"])


@pytest.mark.gpu
def test_evaluation_from_files_end_to_end(ds):
    d, c = ds
    ev = Evaluation(5e-3, 128, 32, 0.95, 0.95, c["dataset_path"], c["model_path"], 128, "std", artifact_dir=d)
    assert (ev.pc_in, ev.pc_p) == (24, 24)
    assert ev.computeOnlyOnce(0) == 0
    assert ev.indice == c["N"] and (ev.grid_shape_y, ev.grid_shape_x) == (138, 300)
    t = _tables(c)
    om = oracle_model(c["model"])
    for time in (0, 2):
        res = ev.timeStep(0, time, False, False, False, False)
        cells = c["sim"][0, time, :c["N"]]                                     # float32, like the file
        grid, dU, dPprev, U = orc.evaluator_grid_deltas(cells, t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct, cases.DATASET_MAXS)
        # interpolation + scatter on the GPU (float64) against the oracle's NumPy statements
        assert np.abs(ev.grid - grid).max() <= 1e-12 * np.abs(grid).max()
        np.testing.assert_allclose(ev.deltaU_change_grid, dU, rtol=0, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(ev.deltaP_prev_grid, dPprev, rtol=0, atol=1e-12, equal_nan=True)
        om.out_scale = cases.DATASET_MAXS[3] * U ** 2
        sol = orc.solve_grid(grid[..., :3], om)
        assert np.abs(res - sol.fields[..., 0]).max() <= 1e-4 * np.abs(sol.fields).max()
        assert np.abs(ev.cfd_results - grid[..., 3] * cases.DATASET_MAXS[3] * pow(np.float32(U), 2.0)).max() <= 1e-12
    # the driver with the reference's argument list (SM_call.py:778): same frames, error summary over the flow cells
    from psm_amd import call_SM_main
    rep = call_SM_main(5e-3, c["model_path"], 128, 0.25, 0.95, 0.95, 128, c["dataset_path"], False, "std", False, False, False,
                       False, 1, 3, artifact_dir=d)
    assert len(rep["sims"]) == 1 and set(rep["overall"]) == {"BIAS", "RMSE", "STDE", "BIAS_block", "RSME_block", "STDE_block"}
    # SM_call.py:553-557, 824-826: the block-level error of the last frame against the oracle's restatement of
    # utils.compute_in_block_error on ITS decoded blocks and de-meaned label blocks
    lay = orc.block_layout("deltas", grid.shape[0], grid.shape[1])
    yb = orc.label_blocks(grid[..., :3], grid[..., 3], lay, 3)
    scale = cases.DATASET_MAXS[3] * float(U) ** 2
    a, b = orc.compute_in_block_error(sol.block_pred, yb * scale, sol.x_blocks[..., 2:3] != 0)
    assert abs(ev.pred_minus_true_block[-1] - a) <= 2e-4 * np.sqrt(b) and abs(ev.pred_minus_true_squared_block[-1] - b) <= 2e-4 * b
    assert rep["overall"]["RSME_block"] > 0 and len(ev.pred_minus_true_block) == len(ev.pred_minus_true)
    flow = ~ev.no_flow_bool
    diff = (res - ev.cfd_results)[flow]
    norm = ev.cfd_results[flow].max() - ev.cfd_results[flow].min()
    assert np.isclose(ev.pred_minus_true[-1], diff.mean() / norm) and rep["overall"]["RMSE"] > 0
    # a frame whose velocity hardly changed is skipped like SM_call.py:413-421
    sim2 = c["sim"].copy(); sim2[0, 1, :c["N"], 5:7] *= 1e-7
    import h5write
    p2 = os.path.join(d, "still.hdf5")
    tb, ob = formats.read_dataset(c["dataset_path"], 0, 0)[1:]
    h5write.write_h5(p2, {"sim_data": sim2, "top_bound": np.repeat(tb, 3, axis=1), "obst_bound": np.repeat(ob, 3, axis=1)})
    ev.dataset_path = p2
    assert isinstance(ev.timeStep(0, 1, False, False, False, False), int)


# ---------------------------------------------------------------------------------------------
# U_to_gradP evaluator (Eval_dual_Dense_onlycil.py)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gds(tmp_path_factory):
    d = tmp_path_factory.mktemp("artefacts_gradp")
    return str(d), cases.build_gradp_dataset_case(str(d))


def _gradp_tables(c):
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    cells = np.asarray(c["sim"][0, 0, :c["N"]], np.float64)
    t = geometry.build_geometry_evaluator(cells[:, 3:5], cells[:, 5], f32(c["top"]), f32(c["obst"]), 5e-3, every=2,
                                          round_digits=2, box="top")
    top32 = np.asarray(c["top"], np.float32)
    return t, (np.min(top32[:, 0]), np.max(top32[:, 0]), np.min(top32[:, 1]), np.max(top32[:, 1]))


def test_oracle_gradp_front_end_matches_reference_run(gds):
    d, c = gds
    gold = cases.load_golden("evaluator_grid_gradp_320x300")
    t, ext = _gradp_tables(c)
    assert (t.ny, t.nx) == (320, 300) and (t.sdfunct[200] == 0).sum() > 10       # the obstacle crosses the hard-wired row
    g, U = orc.evaluator_grid_gradp(c["sim"][0, 1, :c["N"]], t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct, cases.GRADP_MAXS, *ext)
    assert U == float(gold["U_max_norm"])
    np.testing.assert_array_equal(g[170:230, 60:140], gold["grid_crop"])
    np.testing.assert_allclose(g.sum(axis=(0, 1)), gold["grid_sum"], rtol=1e-13)
    np.testing.assert_allclose(np.abs(g).sum(axis=(0, 1)), gold["grid_abs_sum"], rtol=1e-13)


@pytest.mark.gpu
def test_gradp_evaluation_from_files_end_to_end(gds):
    """EvaluationGradP(delta, shape, avance, var_p, var_in, hdf5_path, model_path, max_number_PC): files -> geometry ->
    6-channel grid -> surrogate (both gradient fields) -> four-quadrant integration, against the oracle chain."""
    from psm_amd import EvaluationGradP
    d, c = gds
    ev = EvaluationGradP(5e-3, 128, 96, 0.95, 0.95, c["dataset_path"], c["model_path"], 128, artifact_dir=d)
    assert (ev.pc_in, ev.pc_p) == (24, 24) and ev.artifacts.scaler_kind == "max_abs"
    assert ev.computeOnlyOnce(0) == 0
    t, ext = _gradp_tables(c)
    assert (ev.grid_shape_y, ev.grid_shape_x) == (320, 300)
    p = ev.timeStep(0, 1, False, False, False, False)
    grid, U = orc.evaluator_grid_gradp(c["sim"][0, 1, :c["N"]], t.vtx_m2g, t.wts_m2g, t.indices, t.sdfunct, cases.GRADP_MAXS, *ext)
    assert np.abs(ev.grid - grid).max() <= 1e-12 * np.abs(grid).max()
    sol = orc.solve_grid(grid[..., :3], oracle_model(c["model"]))
    assert np.abs(ev.gradP - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    xl = np.linspace(ext[0], ext[1], 300); yl = np.linspace(ext[2], ext[3], 320)
    cx, cy = orc.integration_center(t.sdfunct, ext[0], ext[1], t.x0, 5e-3)
    assert (cx, cy) == (ev.center_p_x, ev.center_p_y)
    ref = orc.integrate_gradp(sol.fields, t.sdfunct, np.diff(xl)[0], np.diff(yl)[0], cy, cx)
    assert np.abs(p - ref).max() <= 2e-4 * np.abs(ref).max()


# ---------------------------------------------------------------------------------------------
# pressureSM_Poisson evaluator
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_poisson_evaluation_from_files_end_to_end(tmp_path):
    """EvaluationPoisson(..., k, phis_fn) from files: interpolation of the raw columns (NaNs of interpolate_fill kept),
    feature image, 4-channel surrogate (mask = channel 3), assembly, deltaU-change weighting (SM_call.py:519-848)."""
    import scipy.ndimage as ndi
    from psm_amd import EvaluationPoisson
    d = str(tmp_path)
    c = cases.build_dataset_case(d, poisson=True)
    ev = EvaluationPoisson(5e-3, 128, 32, 0.95, 0.95, c["dataset_path"], c["model_path"], 128, "std", 0.5, None, artifact_dir=d)
    assert ev.artifacts.c_in == 4 and ev.artifacts.sdf_ch == 3 and ev.max_abs_delta_p == cases.POISSON_MAXS[4]
    assert ev.computeOnlyOnce(0) == 0
    got = ev.timeStep(0, 1, False, False, False, False, 0.16)
    t = _tables(c)
    cells = c["sim"][0, 1, :c["N"]]
    idx = tuple(t.indices.T)

    def to_grid(v):
        g = np.zeros((t.ny, t.nx))
        g[idx] = orc.interpolate_fill(np.asarray(v).reshape(-1), t.vtx_m2g, t.wts_m2g)
        return g
    dU, dUp = cells[:, 5:7], cells[:, 8:10]
    changed = np.abs(dU - dUp).sum(axis=-1); changed = changed / changed.max()
    U = float(np.max(np.sqrt(np.square(cells[:, 0:1]) + np.square(cells[:, 1:2]))))
    grid, _ = orc.poisson_features(to_grid(cells[:, 0]), to_grid(cells[:, 1]), to_grid(dU[:, 0]), to_grid(dU[:, 1]), t.sdfunct,
                                   0.16, U, 0.5, cases.POISSON_MAXS[:4])
    om = oracle_model(c["model"])
    om.out_scale = cases.POISSON_MAXS[4] * U ** 2
    sol = orc.solve_grid(grid, om)
    prev, w = to_grid(cells[:, 10]), ndi.gaussian_filter(to_grid(changed), sigma=(50, 50), order=0)
    want = prev + ndi.gaussian_filter((sol.fields[..., 0] - prev) * w, sigma=(10, 10), order=0)
    ok = ~np.isnan(want)
    assert ok.mean() > 0.1 and np.array_equal(np.isnan(got), ~ok)
    assert np.abs(got[ok] - want[ok]).max() <= 2e-4 * max(np.abs(want[ok]).max(), np.abs(sol.fields).max())


# ---------------------------------------------------------------------------------------------
# error summaries and mains (Eval_dual_Dense_onlycil.py:667-744; pressureSM_Poisson/SM_call.py:962-1170)
# ---------------------------------------------------------------------------------------------
def _ref_metrics(pred, truth, no_flow):
    """The reference's statements (Eval_dual_Dense_onlycil.py:667-676) on plain arrays."""
    true_mask, pred_mask = truth[~no_flow], pred[~no_flow]
    norm = np.max(true_mask) - np.min(true_mask)
    mask_nan = ~np.isnan(pred_mask - true_mask)
    BIAS_norm = np.mean((pred_mask - true_mask)[mask_nan]) / norm * 100
    RMSE_norm = np.sqrt(np.mean((pred_mask - true_mask)[mask_nan] ** 2)) / norm * 100
    return norm, BIAS_norm, np.sqrt(RMSE_norm ** 2 - BIAS_norm ** 2), RMSE_norm, \
        np.mean((pred_mask - true_mask)[mask_nan]) / norm, np.mean((pred_mask - true_mask)[mask_nan] ** 2) / norm ** 2


def test_error_metrics_are_the_reference_statements():
    from psm_amd import error_metrics
    rng = np.random.default_rng(4)
    truth, pred = rng.standard_normal((40, 60)), rng.standard_normal((40, 60))
    pred[3, 4] = np.nan
    no_flow = rng.random((40, 60)) < 0.2
    m = error_metrics(pred, truth, no_flow)
    norm, b, s, r, e1, e2 = _ref_metrics(pred, truth, no_flow)
    assert (m["normVal"], m["biasNorm"], m["stdeNorm"], m["rmseNorm"], m["mean_err"], m["mean_sq_err"]) == pytest.approx((norm, b, s, r, e1, e2), rel=1e-13)


@pytest.mark.gpu
def test_gradp_main_and_metrics(gds, capsys):
    """main() of the U_to_gradP evaluator with its own argument names: per-frame metrics as the reference computes them
    (against grid channel 3, as written at :667) and the per-simulation BIAS / RMSE / STDE it prints."""
    from psm_amd import EvaluationGradP, main_gradP
    d, c = gds
    out = main_gradP(delta=5e-3, model_directory=c["model_path"], shape=128, var_p=0.95, var_in=0.95, max_number_PC=128,
                     hdf5_path=c["dataset_path"], plot_intermediate_fields=False, save_plots=False, n_ts=2, artifact_dir=d)
    printed = capsys.readouterr().out
    assert "Metrics for the whole simulation:" in printed and "BIAS for the sim: " in printed
    ev = EvaluationGradP(5e-3, 128, 96, 0.95, 0.95, c["dataset_path"], c["model_path"], 128, artifact_dir=d)
    ev.computeOnlyOnce(0)
    e1, e2 = [], []
    for t in range(2):
        field = ev.timeStep(0, t, False, False, False, False)
        norm, b, s, r, m1, m2 = _ref_metrics(field, ev.grid[..., 3], ev.grid[..., 2] == 0)
        fr = out["frames"][t]["reference"]
        assert (fr["normVal"], fr["biasNorm"], fr["rmseNorm"]) == pytest.approx((norm, b, r), rel=1e-6)
        assert ev.last_metrics["p"]["rmseNorm"] == pytest.approx(_ref_metrics(field, ev.grid[..., 5], ev.grid[..., 2] == 0)[3], rel=1e-6)
        e1.append(m1); e2.append(m2)
    BIAS_value = np.mean(e1) * 100
    RMSE_value = np.sqrt(np.mean(e2)) * 100
    assert out["sims"][0]["BIAS"] == pytest.approx(BIAS_value, rel=1e-6) and out["sims"][0]["RMSE"] == pytest.approx(RMSE_value, rel=1e-6)
    assert out["sims"][0]["STDE"] == pytest.approx(np.sqrt(RMSE_value ** 2 - BIAS_value ** 2), rel=1e-5)
    assert ev.pred_minus_true == pytest.approx(e1, rel=1e-6)


@pytest.mark.gpu
def test_poisson_main_and_metrics(tmp_path):
    """call_SM_main of pressureSM_Poisson with its argument list: the three error blocks per frame (delta-p with the
    weighting, delta-p without, p) and the summaries."""
    from psm_amd import EvaluationPoisson, call_SM_main_Poisson
    d = str(tmp_path)
    c = cases.build_dataset_case(d, poisson=True)
    phis = os.path.join(d, "phis.txt")
    np.savetxt(phis, np.array([0.16, 0.2]))
    out = call_SM_main_Poisson(5e-3, c["model_path"], 128, 0.25, 0.95, 0.95, 128, c["dataset_path"], False, "std", 0.5, False, False,
                               False, False, 1, 2, phis, artifact_dir=d, sim_offset=0, time_offset=1)
    ev = EvaluationPoisson(5e-3, 128, 32, 0.95, 0.95, c["dataset_path"], c["model_path"], 128, "std", 0.5, phis, artifact_dir=d)
    ev.computeOnlyOnce(0)
    acc = {k: ([], []) for k in ("", "_deltap_crude", "_p")}
    for t in (1, 2):
        field = ev.timeStep(0, t, False, False, False, False, 0.16)
        U = ev.U_max_norm
        cells = c["sim"][0, t, :c["N"]]
        tb = _tables(c)
        g = np.zeros((tb.ny, tb.nx)); g[tuple(tb.indices.T)] = orc.interpolate_fill(np.asarray(cells[:, 7], np.float64), tb.vtx_m2g, tb.wts_m2g)
        cfd = np.nan_to_num(g / U ** 2) / cases.POISSON_MAXS[4] * cases.POISSON_MAXS[4] * U ** 2
        assert np.abs(ev.cfd_results - cfd).max() <= 1e-12 * np.abs(cfd).max()
        pg = np.zeros((tb.ny, tb.nx)); pg[tuple(tb.indices.T)] = orc.interpolate_fill(np.asarray(cells[:, 2], np.float64), tb.vtx_m2g, tb.wts_m2g)
        pg = np.nan_to_num(pg)
        no_flow = np.nan_to_num(tb.sdfunct) == 0
        for sfx, pred, truth in (("", field, cfd), ("_deltap_crude", ev.deltap_res, cfd), ("_p", (pg - cfd) + field, pg)):
            norm, b, s, r, m1, m2 = _ref_metrics(pred, truth, no_flow)
            lm = ev.last_metrics[sfx.lstrip("_") or "delta_p"]
            assert (lm["normVal"], lm["biasNorm"], lm["rmseNorm"]) == pytest.approx((norm, b, r), rel=1e-6, abs=1e-9), sfx
            acc[sfx][0].append(m1); acc[sfx][1].append(m2)
    for key, sfx in (("delta_p", ""), ("delta_p_no_weighting", "_deltap_crude"), ("p", "_p")):
        BIAS, RMSE = np.mean(acc[sfx][0]) * 100, np.sqrt(np.mean(acc[sfx][1])) * 100
        assert out["overall"][key]["BIAS"] == pytest.approx(BIAS, rel=1e-5, abs=1e-9) and out["overall"][key]["RMSE"] == pytest.approx(RMSE, rel=1e-5)
    assert out["sims"][0]["sim"] == 0 and out["sims"][0]["phi"] == 0.16


_REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(_REF), reason="the reference checkout is only present in the build container")
def test_keras_reader_reads_every_h5_file_of_the_reference():
    """Every Keras HDF5 file the reference ships (56: the Chapter-4 / Chapter-5 surrogates, full-model files with optimizer
    state, the Chapter-3 PINNs) goes through formats.read_keras_dense_weights: a chain of Dense layers whose shapes fit each
    other, finite float32 values.  Reads the files as DATA where they lie (nothing is copied); skipped where the checkout is
    absent (the GPU box)."""
    import glob
    files = sorted(glob.glob(os.path.join(_REF, "**", "*.h5"), recursive=True))
    assert len(files) >= 40
    shapes = set()
    for f in files:
        layers = formats.read_keras_dense_weights(f)
        assert len(layers) >= 2, f
        for (W, b), (W2, _) in zip(layers[:-1], layers[1:]):
            assert W.shape[1] == b.shape[0] == W2.shape[0], f
        assert all(W.dtype == np.float32 and np.isfinite(W).all() and np.isfinite(b).all() for W, b in layers), f
        shapes.add((layers[0][0].shape[0], layers[-1][0].shape[1], len(layers)))
    assert (45, 48, 4) in shapes and (116, 39, 4) in shapes and (2, 5, 8) in shapes     # test_case/weights.h5, M_fU/model_first_.h5, a PINN


@pytest.mark.skipif(not os.path.isdir(_REF), reason="the reference checkout is only present in the build container")
def test_maxs_reader_reads_every_maxs_file_of_the_reference():
    """The reference's normalisation constants (`maxs`: one value per line; `maxs_PCA`: two) -- the values SURVEY.md section 8(c)
    quotes for the shipped solver case and the Chapter-4 evaluator."""
    import glob
    files = sorted(glob.glob(os.path.join(_REF, "**", "maxs*"), recursive=True))
    assert len(files) >= 6
    for f in files:
        v = formats.read_maxs(f)
        assert v.ndim == 1 and np.isfinite(v).all() and (v > 0).all(), f
        assert v.size in (2, 3, 4), f                 # (the Chapter-4 M_fU evaluator's maxs_PCA is a copy of its 3-value maxs)
    np.testing.assert_allclose(formats.read_maxs(os.path.join(_REF, "Thesis_Work/Chapter5/parallelized/test_case/maxs")),
                               [1.0, 0.536133, 0.999023, 0.510742], rtol=1e-5)
    np.testing.assert_allclose(formats.read_maxs(os.path.join(_REF, "Thesis_Work/Chapter5/parallelized/test_case/maxs_PCA")),
                               [147.2389, 26.7201], rtol=1e-5)
