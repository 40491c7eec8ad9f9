"""Host logic of the C-ABI library against the oracle (no GPU): block layouts,
owner map, and the strip-table + offset-chain + paste formulation that the
device kernels execute (replayed on the host by psm_debug_reassemble_host)."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import _lib, surrogate, synthetic

SIZES = [(128, 128), (128, 300), (256, 256), (272, 288), (300, 300), (300, 420), (400, 1000), (512, 512), (131, 257)]


@pytest.mark.parametrize("variant", orc.VARIANTS)
@pytest.mark.parametrize("ny,nx", SIZES)
def test_layout_matches_oracle(variant, ny, nx):
    lay = orc.block_layout(variant, ny, nx)
    blocks, n_x, n_y = surrogate.layout(variant, ny, nx)
    assert (n_x, n_y) == (lay.n_x, lay.n_y)
    assert blocks[:, :2].tolist() == [list(o) for o in lay.origins]
    assert blocks[:, 2:].tolist() == [list(t) for t in lay.tags]


def test_block_counts_of_the_baseline_configs():
    # chapter5 128^2 -> 4, 400x3000 -> 104; deltas 256^2 -> 9, 512^2 -> 30; gradp 256^2 -> 30
    assert len(surrogate.layout("chapter5", 128, 128)[0]) == 4
    assert len(surrogate.layout("chapter5", 400, 3000)[0]) == 104
    assert len(surrogate.layout("deltas", 256, 256)[0]) == 9
    assert len(surrogate.layout("deltas", 512, 512)[0]) == 30
    assert len(surrogate.layout("gradp", 256, 256)[0]) == 30


def test_shipped_case_mesh_gives_the_reference_grid():
    """bench.py's shipped_case leg: the synthetic channel mesh is stretched so that init_func's uniform grid (python_module.py:190-217:
    num = int(round(extent / delta)) from the cell-centre extents) is the reference's 400 x 3000 at delta 0.005 -- through the
    library's host-only shape entry (no GPU)."""
    import ctypes as C
    from psm_amd import _lib, synthetic
    array, top, obst = synthetic.shipped_case_mesh()
    a = np.ascontiguousarray(array, np.float64)
    ny, nx, bd = C.c_int32(), C.c_int32(), np.zeros(4)
    f64 = C.POINTER(C.c_double)
    assert _lib.load().psm_geometry_shape(a.ctypes.data_as(f64), a.shape[0], 0.005, C.byref(ny), C.byref(nx), bd.ctypes.data_as(f64)) == 0
    assert (ny.value, nx.value) == (400, 3000) and 20000 < a.shape[0] < 250000
    assert len(surrogate.layout("chapter5", ny.value, nx.value)[0]) == 104
    assert top[:, 0].min() >= array[:, 2].min() - 0.05 and obst.shape[1] == 2


def test_layout_errors():
    with pytest.raises(_lib.PsmError):
        surrogate.layout("deltas", 100, 300)            # smaller than a block
    with pytest.raises(_lib.PsmError):
        surrogate.owner_map("deltas", 256, 128)         # one block column: reference reads an undefined block
    with pytest.raises(_lib.PsmError):
        surrogate.owner_map("deltas", 512, 512, strict=True)   # p_i == 0: the reference raises


def _blocks_pred(variant, grid, seed):
    """Smooth field + per-block random offsets/noise, cut into the layout's blocks."""
    rng = np.random.default_rng(seed)
    ny, nx = grid.shape[:2]
    lay = orc.block_layout(variant, ny, nx)
    c_out = 2 if variant == "gradp" else 1
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    base = np.stack([np.sin(xx / 37.0 + f) * np.cos(yy / 23.0) + 0.002 * xx for f in range(c_out)], -1)
    bp = orc.extract_blocks(base, lay, c_out).copy()
    bp += rng.standard_normal((lay.B, 1, 1, c_out)) * 0.5
    bp += rng.standard_normal(bp.shape) * 0.01
    return lay, bp


def _oracle_assemble(variant, bp, xb, lay, degenerate="skip"):
    if variant == "chapter5":
        return [orc.assemble_chapter5(bp[..., 0], xb, lay)]
    if variant == "deltas":
        return [orc.assemble_deltas(bp[..., 0], xb, lay, degenerate=degenerate)]
    return [orc.assemble_gradp(w, bp[..., c], xb, lay, degenerate=degenerate) for c, w in enumerate(("dp_dx", "dp_dy"))]


@pytest.mark.parametrize("name", list(cases.GOLDEN_CASES))
def test_host_replay_of_device_reassembly_matches_oracle(name):
    grid, model = cases.build(name)
    v = model.variant
    g3 = np.ascontiguousarray(grid[..., :3], np.float32)
    lay, bp = _blocks_pred(v, g3, seed=5)
    bp32 = bp.astype(np.float32)
    xb = orc.extract_blocks(g3.astype(np.float64), lay, 3)
    fields, offs, shifts = surrogate.debug_reassemble_host(v, g3, bp32, bp.shape[-1])
    for c, a in enumerate(_oracle_assemble(v, bp32.astype(np.float64), xb, lay)):
        assert a.covered.all()
        np.testing.assert_allclose(offs[c], a.offsets, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(shifts[c], a.shift, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(fields[..., c], a.field, rtol=0, atol=1e-4)


@pytest.mark.parametrize("variant,ny,nx", [("gradp", 256, 256), ("deltas", 512, 512), ("gradp", 384, 512)])
def test_degenerate_last_row_is_skipped(variant, ny, nx):
    """p_i == 0: the reference is undefined (NaN field / broadcast error); the build
    leaves the duplicate last block row out."""
    grid = np.ascontiguousarray(synthetic.channel_grid(ny, nx, seed=3), np.float32)
    lay, bp = _blocks_pred(variant, grid, seed=9)
    bp32 = bp.astype(np.float32)
    xb = orc.extract_blocks(grid.astype(np.float64), lay, 3)
    fields, offs, _ = surrogate.debug_reassemble_host(variant, grid, bp32, bp.shape[-1])
    assert np.isfinite(fields).all()
    for c, a in enumerate(_oracle_assemble(variant, bp32.astype(np.float64), xb, lay)):
        assert a.covered.all()
        last = np.array([t[0] == lay.n_y + 1 for t in lay.tags])
        assert np.isnan(offs[c][last]).all() and np.isnan(a.offsets[last]).all()
        np.testing.assert_allclose(fields[..., c], a.field, rtol=0, atol=1e-4)
    own = surrogate.owner_map(variant, ny, nx)
    assert (own >= 0).all()
    assert not np.isin(own // (128 * 128), np.nonzero(last)[0]).any()


def test_strict_gradp_reproduces_numpy_nan_semantics():
    grid = np.ascontiguousarray(synthetic.channel_grid(256, 256, seed=3), np.float32)
    lay, bp = _blocks_pred("gradp", grid, seed=9)
    xb = orc.extract_blocks(grid.astype(np.float64), lay, 3)
    fields, _, _ = surrogate.debug_reassemble_host("gradp", grid, bp.astype(np.float32), 2, strict=True)
    a = orc.assemble_gradp("dp_dx", bp[..., 0], xb, lay, degenerate="strict")
    assert np.isnan(a.field).all() and np.isnan(fields[..., 0]).all()     # UGP:340 -> UGP:359 poisons the field
