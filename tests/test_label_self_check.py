"""a8 + the reference's own check of the assembly algorithm: the labels, de-meaned per block over the flow cells
(SM_call.py:487-488; Eval_dual_Dense_onlycil.py:509-511), pushed through the same reassembly as the prediction
(SM_call.py:577-580 commented, Eval_dual_Dense_onlycil.py:546-547 live) -- on the GPU, against the output of the
reference's statements (gradp_272x288.npz:label_fields) and against the oracle for the deltas variant."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import Evaluation, EvaluationGradP, GridSurrogate

pytestmark = pytest.mark.gpu


def test_gradp_labels_through_the_gpu_reassembly_match_the_reference_run():
    grid, model = cases.build("gradp_272x288")
    gold = cases.load_golden("gradp_272x288")["label_fields"]
    ev = EvaluationGradP(5e-3, 128, 96, 0.95, 0.995, None, None, 32, model=model)
    got = ev.label_self_check(grid[..., :3], grid[..., 3:5])
    assert got.shape == gold.shape and np.isfinite(got).all()
    assert np.abs(got - gold).max() <= 2e-4 * np.abs(gold).max()
    # the de-meaned label blocks themselves (float64 in the reference, float32 here)
    lay = orc.block_layout("gradp", grid.shape[0], grid.shape[1])
    xb = orc.extract_blocks(grid, lay, 3)
    yb = orc.extract_blocks(grid[..., 3:5], lay, 2).copy()
    for b in range(lay.B):
        m = xb[b, :, :, 2] != 0
        for ch in range(2):
            yb[b, :, :, ch][m] -= np.mean(yb[b, :, :, ch][m])
    assert np.abs(ev.y_array - yb).max() <= 2e-6 * np.abs(yb).max()


def test_deltas_labels_and_empty_blocks():
    grid, model = cases.build("deltas_nan_256x256")          # holds a solid band: blocks with few / no flow cells
    labels = grid[..., 3]
    ev = Evaluation(5e-3, 128, 32, 0.95, 0.995, None, None, 32, "max_abs", model=model)
    got = ev.label_self_check(grid[..., :3], labels)[..., 0]
    lay = orc.block_layout("deltas", grid.shape[0], grid.shape[1])
    xb = orc.extract_blocks(grid, lay, 3)
    yb = orc.extract_blocks(grid[..., 3:4], lay, 1).copy()
    for b in range(lay.B):
        m = xb[b, :, :, 2] != 0
        if m.any():
            yb[b, :, :, 0][m] -= np.mean(yb[b, :, :, 0][m])
    ref = orc.assemble_deltas(yb[..., 0], xb, lay, degenerate="strict").field
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max()
    # an all-solid grid: nothing to de-mean, blocks come back unchanged
    solid = np.zeros_like(grid[..., :3])
    with GridSurrogate(model, grid.shape[0], grid.shape[1]) as sur:
        blocks = sur.label_blocks(solid, labels)
    np.testing.assert_array_equal(blocks[..., 0], orc.extract_blocks(grid[..., 3:4], lay, 1)[..., 0].astype(np.float32))
