"""Pin the CPU oracle: it must reproduce the outputs that the reference's own
statements produced (tests/golden/*.npz, written by tests/golden/make_golden.py)
on the same seeded inputs."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc


def oracle_model(m):
    sc = orc.Scaler(m.scaler_kind, m.in_a, m.in_b, m.out_a, m.out_b)
    return orc.Model(m.variant, m.c_in, m.c_out, m.comp_in, m.mean_in, m.comp_out, m.mean_out,
                     m.weights, sc, m.out_scale, m.S, m.ov, m.sdf_ch, getattr(m, "conv1d", ()), getattr(m, "attention", None))


@pytest.mark.parametrize("name", list(cases.GOLDEN_CASES))
def test_oracle_matches_reference_run(name):
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    sol = orc.solve_grid(grid, oracle_model(model), degenerate="strict")
    assert sol.x_blocks.shape[0] == int(gold["n_blocks"])
    # PCA encode + scaler: float64 on both sides (sklearn subtracts mean@comp.T after the product)
    np.testing.assert_allclose(sol.x_input, gold["x_input"], rtol=1e-9, atol=1e-11)
    # reassembled field(s): the MLP is float32 in both runs, the rest float64
    ref = gold["fields"]
    assert not np.isnan(ref).any()
    np.testing.assert_allclose(sol.fields, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
    for a in sol.assemblies:
        assert a.covered.all()


def test_oracle_strict_mode_is_the_reference_on_degenerate_grids():
    """p_i == 0 (BASELINE configs[1] / configs[4] shapes): the reference's own statements give dp_dx = NaN everywhere
    (UGP:340 -> UGP:359), dp_dy = NaN on the rows the last block row pastes and finite elsewhere, and raise a
    broadcast error for deltaU_to_deltaP (SMD:335).  degenerate='strict' must reproduce exactly that."""
    name = "gradp_degenerate_256x256"
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    with np.errstate(all="ignore"):
        sol = orc.solve_grid(grid, oracle_model(model), degenerate="strict")
    ref = gold["fields"]
    assert np.isnan(ref[..., 0]).all() and np.isnan(ref[224:, :, 1]).all() and not np.isnan(ref[:224, :, 1]).any()
    np.testing.assert_array_equal(np.isnan(sol.fields), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(sol.fields[ok], ref[ok], rtol=1e-6, atol=1e-6 * np.abs(ref[ok]).max())
    np.testing.assert_allclose(sol.x_input, gold["x_input"], rtol=1e-9, atol=1e-11)
    name = "deltas_degenerate_512x512"
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    assert int(gold["raised"]) == 1 and int(gold["n_blocks"]) == 30
    with pytest.raises(ValueError):
        orc.solve_grid(grid, oracle_model(model), degenerate="strict")
    # the build-defined default still encodes the reference's block list: same coefficients
    sol = orc.solve_grid(grid, oracle_model(model))
    np.testing.assert_allclose(sol.x_input, gold["x_input"], rtol=1e-9, atol=1e-11)
    assert np.isfinite(sol.fields).all()


def test_labels_through_gradp_reassembly():
    """UGP:509-511,546-547: the labels, de-meaned per block, pushed through the
    same reassembly (the reference's own self-check)."""
    name = "gradp_272x288"
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    lay = orc.block_layout("gradp", grid.shape[0], grid.shape[1])
    xb = orc.extract_blocks(grid, lay, 3)
    yb = orc.extract_blocks(grid[..., 3:5], lay, 2).copy()
    for b in range(lay.B):
        m = xb[b, :, :, 2] != 0
        for ch in range(2):
            yb[b, :, :, ch][m] -= np.mean(yb[b, :, :, ch][m])
    for ch, which in enumerate(("dp_dx", "dp_dy")):
        a = orc.assemble_gradp(which, yb[..., ch], xb, lay, degenerate="strict")
        np.testing.assert_allclose(a.field, gold["label_fields"][..., ch], rtol=1e-10, atol=1e-12)


def test_real_weights_fixture_is_the_reference_file():
    W, maxs, maxs_pca = cases.real_chapter5_weights()
    assert [w.shape for w, _ in W] == [(45, 512), (512, 512), (512, 512), (512, 48)]
    # statistics recorded in SURVEY.md §8c for Thesis_Work/Chapter5/parallelized/test_case/weights.h5
    assert abs(float(W[0][0].min()) + 1.279) < 1e-3 and abs(float(W[0][0].max()) - 0.992) < 1e-3
    np.testing.assert_allclose(maxs, [1.0, 0.536133, 0.999023, 0.510742], rtol=1e-5)
    np.testing.assert_allclose(maxs_pca, [147.2389, 26.7201], rtol=1e-5)


def test_oracle_refuses_a_gradp_grid_whose_first_block_is_solid():
    """UGP:294-300: the inlet reference of dp/dx is taken on the first column of block 0 that holds a flow cell, and the
    reference asserts that there is one ("At least the right-most column ... must belong to the flow domain, or this won't
    work").  The oracle restates the failure as a ValueError instead of running off the block; the GPU library's answer for
    such a grid (NaN dp/dx, never a plausible field) is pinned in tests/test_gpu_parity.py."""
    import psm_amd
    from psm_amd import synthetic
    from bench import oracle_model
    model = synthetic.make_model("gradp", p_in=6, p_out=5)
    g = synthetic.channel_grid(272, 288, seed=5, obstacle="none").astype(np.float64)
    g[:128, :128, :] = 0.0
    with pytest.raises(ValueError, match="first block has no flow cell"):
        orc.solve_grid(g, oracle_model(model))
    g[5, 127, :] = 0.3                                       # one flow cell in the block's last column: defined again
    with np.errstate(all="ignore"):
        assert orc.solve_grid(g, oracle_model(model)).fields.shape == (272, 288, 2)
