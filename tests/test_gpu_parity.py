"""Parity of the HIP path (through the C-ABI) with the oracle and with the golden
vectors produced by the reference's own statements.  Needs a real MI355X.

Tolerances (north_star: "within a stated fp32 tolerance"): the GPU path computes
everything in float32 (exact-f32 MFMA, f32 reductions) where the reference uses
float64 for PCA / reassembly and float32 for the network:
  * PCA coefficients / network input : relative L2 <= 2e-6, max abs <= 2e-5 * max|x|
  * network output, decoded blocks   : relative L2 <= 1e-5
  * per-block offsets, final fields  : max abs <= 1e-4 * max|field|
"""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate, Evaluation, EvaluationGradP, SolverModule, _lib, surrogate, synthetic

pytestmark = pytest.mark.gpu


def oracle_model(m):
    sc = orc.Scaler(m.scaler_kind, m.in_a, m.in_b, m.out_a, m.out_b)
    return orc.Model(m.variant, m.c_in, m.c_out, m.comp_in, m.mean_in, m.comp_out, m.mean_out,
                     m.weights, sc, m.out_scale, m.S, m.ov, m.sdf_ch, getattr(m, "conv1d", ()), getattr(m, "attention", None))


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


def check_against_oracle(sur, grid, model, sol, n_cases=1, case=0):
    B = sur.B
    rows = slice(case * B, (case + 1) * B)
    x = sur.stage("x_input", n_cases)[rows]
    assert rel_l2(x, sol.x_input) <= 2e-6
    assert np.abs(x - sol.x_input).max() <= 2e-5 * np.abs(sol.x_input).max()
    res = sur.stage("res", n_cases)[rows]
    ref_res = oracle_model(model).scaler.inv(sol.res.astype(np.float64))
    assert rel_l2(res, ref_res) <= 1e-5
    bp = sur.stage("block_pred", n_cases)[rows]
    assert rel_l2(bp, sol.block_pred) <= 1e-5
    offs = sur.stage("offsets", n_cases)[case]
    for c, a in enumerate(sol.assemblies):
        scale = max(np.nanmax(np.abs(a.field)), 1e-6)
        np.testing.assert_allclose(offs[c], a.offsets, rtol=0, atol=1e-4 * scale, equal_nan=True)


@pytest.mark.parametrize("name", list(cases.GOLDEN_CASES))
def test_golden_cases(name):
    """HIP path vs the outputs of the reference's own statements (tests/golden)."""
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    sp = cases.GOLDEN_CASES[name]
    out_scale = [model.out_scale] if model.variant == "deltas" else None
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, grid.shape[0], grid.shape[1]) as sur:
        fields = sur.solve(g32, out_scale=out_scale)[0]
        assert sur.B == int(gold["n_blocks"])
        x = sur.stage("x_input")
        sol = orc.solve_grid(g32.astype(np.float64), oracle_model(model), degenerate="strict")
        check_against_oracle(sur, g32, model, sol)
    # the golden x_input was computed from the float64 grid; the f32 rounding of the input is ~6e-8
    assert rel_l2(x, gold["x_input"]) <= 5e-6
    ref = gold["fields"]
    assert np.isfinite(fields).all()
    assert np.abs(fields - ref).max() <= 2e-4 * np.abs(ref).max(), np.abs(fields - ref).max()
    assert rel_l2(fields, ref) <= 5e-5


@pytest.mark.parametrize("bind", [False, True])
def test_strict_degenerate_is_the_reference_on_p_i_zero_grids(bind):
    """strict_degenerate=1 on the GPU: on the 256-row U_to_gradP grid (BASELINE configs[1] shape) the reference's own run
    gives dp_dx = NaN everywhere (UGP:340 -> UGP:359) and dp_dy = NaN on the last block row's paste, finite elsewhere
    (tests/golden/gradp_degenerate_256x256.npz); on a 512-row deltaU_to_deltaP grid (configs[4] shape) it raises
    (SMD:335) -> PSM_ERR_UNSUPPORTED.  General and geometry-bound paths."""
    name = "gradp_degenerate_256x256"
    grid, model = cases.build(name)
    ref = cases.load_golden(name)["fields"]
    g32 = grid[..., :3].astype(np.float32)
    with GridSurrogate(model, 256, 256, strict_degenerate=True) as sur:
        assert sur.B == 30
        if bind:
            assert sur.bind_geometry(g32)
        fields = sur.solve(g32)[0]
        assert sur.geometry_bound == bind
    np.testing.assert_array_equal(np.isnan(fields), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert ok.sum() == 224 * 256
    assert np.abs(fields[ok] - ref[ok]).max() <= 2e-4 * np.abs(ref[ok]).max()
    # the same grid in the build-defined default mode: finite everywhere
    with GridSurrogate(model, 256, 256) as sur:
        if bind:
            assert sur.bind_geometry(g32)
        assert np.isfinite(sur.solve(g32)[0]).all()
    grid, model = cases.build("deltas_degenerate_512x512")
    with pytest.raises(_lib.PsmError) as e:
        GridSurrogate(model, 512, 512, strict_degenerate=True)
    assert e.value.code == -5                           # PSM_ERR_UNSUPPORTED: the reference raises here
    with GridSurrogate(model, 512, 512) as sur:
        g32 = grid[..., :3].astype(np.float32)
        if bind:
            assert sur.bind_geometry(g32)
        f = sur.solve(g32, out_scale=[model.out_scale])[0]
        x = sur.stage("x_input")
    assert np.isfinite(f).all()
    assert rel_l2(x, cases.load_golden("deltas_degenerate_512x512")["x_input"]) <= 5e-6


def test_config1_gradp_256_p128():
    """BASELINE config 1: 256x256 U_to_gradP, batch 1, fp32, P_i = P_o = 128, MLP_small."""
    model = synthetic.make_model("gradp")
    grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        fields = sur.solve(grid)[0]
        assert sur.B == 30
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.isfinite(fields).all()
    assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    print("config1 rel L2 =", rel_l2(fields, sol.fields))


def test_config2_deltas_sequence():
    """BASELINE config 2: 256x256 deltaU_to_deltaP, sequential steps with per-step out_scale."""
    model = synthetic.make_model("deltas")
    om = oracle_model(model)
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.B == 9
        for step in range(4):
            grid = synthetic.delta_grid(256, 256, seed=2, step=step).astype(np.float32)
            sc = 0.51 * (1.0 + 0.1 * step) ** 2
            fields = sur.solve(grid, out_scale=[sc])[0]
            om.out_scale = sc
            sol = orc.solve_grid(grid.astype(np.float64), om)
            check_against_oracle(sur, grid, model, sol)
            assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_submission_ring_equals_synchronous_solves():
    """psm_submit_grid / psm_wait_grid (pinned ring, copies overlapping the kernels): the same fields,
    bit for bit, as the synchronous host entry on a sequence of different time steps with per-step
    out_scale; a full ring is refused, an unknown ticket too."""
    model = synthetic.make_model("deltas")
    grids = [synthetic.delta_grid(256, 256, seed=2, step=s).astype(np.float32) for s in range(10)]
    scales = [0.51 * (1.0 + 0.1 * s) ** 2 for s in range(10)]
    with GridSurrogate(model, 256, 256) as sur:
        ref = [sur.solve(g, out_scale=[sc]) for g, sc in zip(grids, scales)]
        got, pending = [], []
        for g, sc in zip(grids, scales):
            if len(pending) == 8:                       # PSM_RING_SLOTS
                with pytest.raises(_lib.PsmError) as e:
                    sur.submit(g, out_scale=[sc])
                assert e.value.code == -2               # PSM_ERR_STATE
                got.append(sur.wait(pending.pop(0)))
            buf = g.copy()
            pending.append(sur.submit(buf, out_scale=[sc]))
            buf[:] = np.nan                             # the caller's buffer is free on return
        while pending:
            got.append(sur.wait(pending.pop(0)))
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        with pytest.raises(ValueError):
            sur.wait(12345)


def test_config3_case_batch_equals_single_cases():
    """BASELINE config 3 (per-GPU shard): 8 random-obstacle cases in one call."""
    model = synthetic.make_model("deltas")
    grids = synthetic.random_obstacle_cases(8, 256, 256, seed=3).astype(np.float32)
    scales = np.linspace(0.5, 1.2, 8).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=8) as sur:
        batch = sur.solve(grids, out_scale=scales)
        om = oracle_model(model)
        for c in (0, 3, 7):
            om.out_scale = float(scales[c])
            sol = orc.solve_grid(grids[c].astype(np.float64), om)
            check_against_oracle(sur, grids[c], model, sol, n_cases=8, case=c)
            assert np.abs(batch[c] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        singles = np.stack([sur.solve(grids[c], out_scale=[scales[c]])[0] for c in range(8)])
    # 8 cases (96 block rows) and single cases (32 rows) may take different dense-layer paths: same
    # arithmetic, different summation order
    np.testing.assert_allclose(batch, singles, rtol=0, atol=2e-6 * np.abs(singles).max())


def test_config3_all_64_cases_on_one_card():
    """BASELINE config 3 as this pool can run it: the whole batch of 64 random-obstacle cases in ONE call on one card (576 block
    rows: the M-tiled encode, psm_encode_x6_mt_kernel, one slab per K group).  Four of the cases against the oracle at the usual
    tolerance, every case against the same case solved alone (other encode form and dense-layer path: float32 summation order
    only), general path and one bound geometry per case slot."""
    model = synthetic.make_model("deltas")
    n = 64
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=3).astype(np.float32)
    scales = np.linspace(0.5, 1.5, n).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=n) as sur:
        batch = sur.solve(grids, out_scale=scales)
        om = oracle_model(model)
        for c in (0, 21, 42, 63):
            om.out_scale = float(scales[c])
            sol = orc.solve_grid(grids[c].astype(np.float64), om)
            assert np.abs(batch[c] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max(), c
        singles = np.stack([sur.solve(grids[c], out_scale=[scales[c]])[0] for c in range(n)])
        np.testing.assert_allclose(batch, singles, rtol=0, atol=3e-6 * np.abs(singles).max())
        assert sur.bind_geometry(grids)
        bound = sur.solve(grids, out_scale=scales)
        np.testing.assert_allclose(bound, batch, rtol=0, atol=5e-5 * np.abs(batch).max())
        # 50 cases through the same handle: 450 block rows (padded to 480: a last row group of one tile), other K grouping
        part = sur.solve(grids[:50], out_scale=scales[:50])
        np.testing.assert_allclose(part, singles[:50], rtol=0, atol=3e-6 * np.abs(singles).max())


@pytest.mark.parametrize("c_in,p_in,variant", [(4, 48, "deltas"), (3, 40, "chapter5"), (3, 100, "gradp")])
def test_large_batch_encode_other_shapes(c_in, p_in, variant):
    """The M-tiled encode (>= 432 block rows) on shapes the 64-case test does not reach: four input channels (K slices of 256), fewer
    component tiles than waves (40 / 48 components: waves 2, 3 idle), a Chapter-5 layout (16 blocks per 256 x 256 case: 30 cases = 480 rows),
    U_to_gradP's 30 blocks per case (16 cases = 480 rows).  Every case against the same case solved alone, three against the oracle."""
    model = synthetic.make_model(variant, p_in=p_in, p_out=32, c_in=c_in, scaler_kind="min_max" if c_in == 4 else None)
    if c_in == 4:
        model.sdf_ch = 3
    with GridSurrogate(model, 256, 256) as probe:
        blocks = probe.B
    n = -(-432 // blocks) + 1
    g3 = synthetic.random_obstacle_cases(n, 256, 256, seed=21).astype(np.float32)
    grids = np.concatenate([g3[..., :1] * g3[..., 1:2], g3], axis=-1).astype(np.float32) if c_in == 4 else g3
    with GridSurrogate(model, 256, 256, max_cases=n) as sur:
        assert n * sur.B >= 432
        batch = sur.solve(grids)
        singles = np.stack([sur.solve(grids[c])[0] for c in range(n)])
        om = oracle_model(model)
        for c in (0, n // 2, n - 1):
            sol = orc.solve_grid(grids[c].astype(np.float64), om)
            assert np.abs(batch[c] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max(), c
    np.testing.assert_allclose(batch, singles, rtol=0, atol=3e-6 * np.abs(singles).max())


def test_config0_chapter5_real_weights_via_solver_module():
    grid, model = cases.build("chapter5_128x128_real")
    gold = cases.load_golden("chapter5_128x128_real")
    p = SolverModule(model).py_func_grid(grid.astype(np.float32))
    assert np.abs(p - gold["fields"][..., 0]).max() <= 2e-4 * np.abs(gold["fields"]).max()


def test_unaligned_grid_and_four_channels():
    """Odd Nx (scalar-load path of the encode kernel) and C_in = 4 (pressureSM_Poisson features)."""
    model = synthetic.make_model("chapter5", p_in=40, p_out=24)
    grid = synthetic.channel_grid(131, 257, seed=4).astype(np.float32)
    with GridSurrogate(model, 131, 257) as sur:
        f = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.abs(f - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    m4 = synthetic.make_model("deltas", p_in=48, p_out=32, c_in=4, scaler_kind="min_max")
    m4.sdf_ch = 3
    g3 = synthetic.channel_grid(256, 320, seed=5)
    g4 = np.concatenate([g3[..., :1] * g3[..., 1:2], g3], axis=-1).astype(np.float32)
    with GridSurrogate(m4, 256, 320) as sur:
        f = sur.solve(g4)[0]
        sol = orc.solve_grid(g4.astype(np.float64), oracle_model(m4))
        check_against_oracle(sur, g4, m4, sol)
    assert np.abs(f - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


@pytest.mark.parametrize("c_in,ny,nx", [(3, 256, 256), (3, 200, 259), (4, 256, 320)])
def test_single_case_encode_with_two_slices_per_workgroup(c_in, ny, nx):
    """One case, four component tiles (97 ... 128 input components): psm_encode_pair_kernel -- a workgroup takes two K slices and
    half of the components and adds the slices through LDS (128 slabs) -- in its aligned, unaligned (odd Nx) and four-channel
    instances; every stage against the oracle, and more than 32 block rows of the same model through the one-slice-per-slab
    kernels on the same handle."""
    model = synthetic.make_model("deltas", p_in=120, p_out=128, c_in=c_in, scaler_kind="std")
    if c_in == 4:
        model.sdf_ch = 3
    g3 = synthetic.channel_grid(ny, nx, seed=8)
    grid = (g3 if c_in == 3 else np.concatenate([g3[..., :1] * g3[..., 1:2], g3], axis=-1)).astype(np.float32)
    with GridSurrogate(model, ny, nx, max_cases=5) as sur:
        assert sur.B <= 32
        f = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
        assert np.abs(f - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        five = np.stack([grid * np.float32(1.0 + 0.1 * k) for k in range(5)])
        five[..., model.sdf_ch] = grid[..., model.sdf_ch]
        got = sur.solve(five)
        assert np.abs(got[0] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        again = sur.solve(grid)[0]
        np.testing.assert_array_equal(again, f)


def test_big_architecture_and_many_components():
    """MLP_small_unet (9 layers, widths 512..32..512) with 200 input / 136 output PCs (more than 4 N-tiles)."""
    model = synthetic.make_model("deltas", p_in=200, p_out=136, arch="MLP_small_unet", scaler_kind="std")
    grid = synthetic.channel_grid(256, 352, seed=6).astype(np.float32)
    with GridSurrogate(model, 256, 352) as sur:
        f = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.abs(f - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_device_pointer_api_graph_and_eager_agree(monkeypatch):
    from hipmem import DeviceArray
    model = synthetic.make_model("gradp", p_in=64, p_out=64)
    grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        host = sur.solve(grid)[0]
        d_in = DeviceArray(grid)
        d_out = DeviceArray(shape=(256, 256, 2), dtype=np.float32)
        for _ in range(3):      # plain stream launches on the handle's own stream
            sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        np.testing.assert_array_equal(d_out.numpy(), host)
        ms = sur.profile(d_in.ptr, 1, d_out.ptr)       # launches with events around every kernel group
        np.testing.assert_array_equal(d_out.numpy(), host)
        assert ms["encode"] > 0 and ms["decode"] > 0 and ms["paste"] > 0
        sur.enable_kernel_timing("encode")
        sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        total, n = sur.kernel_timing("encode")
        sur.enable_kernel_timing("encode", False)
        assert n == 1 and total > 0
        assert sur.event_pair_overhead_ms(50) >= 0
    monkeypatch.setenv("PSM_GRAPH", "1")          # a second handle that replays a captured hipGraph
    with GridSurrogate(model, 256, 256) as sur:
        for _ in range(3):      # first call captures the graph, the others replay it
            sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        np.testing.assert_array_equal(d_out.numpy(), host)
    d_in.free(); d_out.free()


def test_reassembly_properties_full_size():
    """a12 at the BASELINE grid sizes (incl. the shipped 400x3000 case, 104 blocks):
    (i)   per-block constants are absorbed by the offsets (field unchanged),
    (ii)  the device result equals the oracle's reassembly of the same blocks,
    (iii) a global field cut into blocks comes back up to one constant (the reference's
          own self-check, UGP:546-547 / SMD:577-580) -- exact for gradp without an
          obstacle; elsewhere the reference compares strips of different cells
          (SMD:292 vs SMD:283, masks of SMD:235) and is only "almost perfect"."""
    rng = np.random.default_rng(0)
    for variant, ny, nx in (("gradp", 256, 256), ("deltas", 256, 256), ("deltas", 512, 512), ("chapter5", 400, 3000)):
        model = synthetic.make_model(variant, p_in=8, p_out=8)
        c_out = model.c_out
        grid = synthetic.channel_grid(ny, nx, seed=8).astype(np.float32)
        yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        truth = np.stack([np.sin(xx / 50.0 + c) + np.cos(yy / 31.0) for c in range(c_out)], -1)
        lay = orc.block_layout(variant, ny, nx)
        bp = orc.extract_blocks(truth, lay, c_out).astype(np.float32)
        open_grid = synthetic.channel_grid(ny, nx, seed=8, obstacle="none").astype(np.float32)
        with GridSurrogate(model, ny, nx) as sur:
            f0 = sur.reassemble(grid, bp)
            f1 = sur.reassemble(grid, bp + rng.standard_normal((lay.B, 1, 1, c_out)).astype(np.float32))
            f2 = sur.reassemble(open_grid, bp)
        assert np.abs(f0 - f1).max() <= 2e-4
        xb = orc.extract_blocks(grid.astype(np.float64), lay, 3)
        for c in range(c_out):
            if variant == "gradp":
                ref = orc.assemble_gradp(("dp_dx", "dp_dy")[c], bp[..., c], xb, lay).field
                d = f2[..., c] - truth[..., c]
                assert d.max() - d.min() <= 2e-4
            elif variant == "deltas":
                ref = orc.assemble_deltas(bp[..., c], xb, lay).field
            else:
                ref = orc.assemble_chapter5(bp[..., c], xb, lay).field
            assert np.abs(f0[..., c] - ref).max() <= 2e-4


def test_full_solve_on_large_grids():
    """The launch paths the 256x256 configs do not reach, against the oracle: gradp 512x512 (182 blocks: more than
    128 block rows -> un-fused reduce, 32-row dense tiles, chunked encode/decode, separate chain + paste launches)
    and the shipped 400x3000 Chapter-5 shape (104 blocks)."""
    for variant, ny, nx, B in (("gradp", 512, 512, 182), ("chapter5", 400, 3000, 104)):
        model = synthetic.make_model(variant, p_in=16, p_out=24, seed_pca=90 + ny, seed_w=4)
        grid = synthetic.channel_grid(ny, nx, seed=6, extra_channels=0).astype(np.float32)
        with GridSurrogate(model, ny, nx) as sur:
            assert sur.B == B
            fields = sur.solve(grid)[0]
            sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
            check_against_oracle(sur, grid, model, sol)
        assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_more_than_64_block_columns_uses_the_serial_chain():
    """gradp strip 160 x 2240: 67 block columns (> one wavefront) -> the single-lane form of the offset chain
    (psm_chain_v, the host replay's code) inside the chain kernel."""
    ny, nx = 160, 2240
    model = synthetic.make_model("gradp", p_in=8, p_out=8)
    lay = orc.block_layout("gradp", ny, nx)
    assert lay.n_x + 1 > 64
    grid = synthetic.channel_grid(ny, nx, seed=12).astype(np.float32)
    rng = np.random.default_rng(1)
    bp = rng.standard_normal((lay.B, 128, 128, 2)).astype(np.float32)
    xb = orc.extract_blocks(grid.astype(np.float64), lay, 3)
    with GridSurrogate(model, ny, nx) as sur:
        f = sur.reassemble(grid, bp)
    for c, which in enumerate(("dp_dx", "dp_dy")):
        ref = orc.assemble_gradp(which, bp[..., c], xb, lay).field
        assert np.abs(f[..., c] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_reference_shaped_assemble_prediction():
    """Evaluation.assemble_prediction with the reference's argument list (SM_call.py:182,
    Eval_dual_Dense_onlycil.py:255) on label-like blocks, against the golden label fields."""
    name = "gradp_272x288"
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    lay = orc.block_layout("gradp", 272, 288)
    xb = orc.extract_blocks(grid, lay, 3)
    yb = orc.extract_blocks(grid[..., 3:5], lay, 2).copy()
    for b in range(lay.B):
        m = xb[b, :, :, 2] != 0
        for ch in range(2):
            yb[b, :, :, ch][m] -= np.mean(yb[b, :, :, ch][m])
    ev = EvaluationGradP(5e-3, 128, 96, 0.95, 0.95, None, None, 512, model=model)
    ev.x_array = xb
    idx = [list(t) for t in lay.tags]
    for ch, which in enumerate(("dp_dx", "dp_dy")):
        r = ev.assemble_prediction(which, yb[..., ch], idx, lay.n_x, lay.n_y, False, 288, 272)
        assert r.shape == (1, 272, 288, 1)
        assert np.abs(r[0, :, :, 0] - gold["label_fields"][..., ch]).max() <= 1e-4


def test_errors_are_reported_not_fatal():
    model = synthetic.make_model("deltas", p_in=8, p_out=8)
    with pytest.raises(_lib.PsmError) as e:
        GridSurrogate(model, 256, 128)                 # single block column: reference undefined
    assert e.value.code == -5
    with GridSurrogate(model, 256, 256, max_cases=2) as sur:
        with pytest.raises(_lib.PsmError):
            sur.solve(np.zeros((3, 256, 256, 3), np.float32))   # more cases than max_cases
        with pytest.raises(ValueError):
            sur.solve(np.zeros((1, 200, 256, 3), np.float32))
    bad = synthetic.make_model("deltas", p_in=8, p_out=8)
    bad.scaler_kind = "zscore"
    with pytest.raises(ValueError, match="Standardization method not valid"):
        GridSurrogate(bad, 256, 256)


def test_config4_bf16_512():
    """BASELINE config 4: 512x512, bf16 operands / f32 accumulation (v_mfma_f32_32x32x16_bf16).
    No reference counterpart (the reference is float64/float32): compared with the oracle's bf16
    emulation (same roundings, exact sums) and, loosely, with the full-precision oracle.
    Tolerance: bf16 has 8 significant bits; an activation that sits on a rounding boundary may
    round differently after float32 vs float64 accumulation."""
    model = synthetic.make_model("deltas")
    grid = synthetic.channel_grid(512, 512, seed=4, noise=0.05).astype(np.float32)
    with GridSurrogate(model, 512, 512, precision="bf16") as sur:
        assert sur.B == 30
        f = sur.solve(grid)[0]
        x = sur.stage("x_input")
        bp = sur.stage("block_pred")
    om = oracle_model(model)
    emu = orc.solve_grid(grid.astype(np.float64), om, precision="bf16")
    full = orc.solve_grid(grid.astype(np.float64), om)
    assert np.isfinite(f).all()
    e_x, e_bp, e_f = rel_l2(x, emu.x_input), rel_l2(bp, emu.block_pred), rel_l2(f, emu.fields)
    print("bf16 vs emulation: x_in %.2e block_pred %.2e fields %.2e; vs f64 oracle fields %.2e" % (e_x, e_bp, e_f, rel_l2(f, full.fields)))
    assert e_x <= 1e-4            # same roundings, f32 vs f64 accumulation only
    assert e_bp <= 5e-3 and e_f <= 5e-3     # a coefficient on a bf16 rounding boundary may flip (f32 vs f64 sums)
    assert rel_l2(x, full.x_input) <= 2e-2 and rel_l2(f, full.fields) <= 5e-2


def test_bf16_all_variants_small():
    for variant, ny, nx in (("gradp", 272, 288), ("chapter5", 300, 400)):
        model = synthetic.make_model(variant, p_in=40, p_out=24)
        grid = synthetic.channel_grid(ny, nx, seed=9).astype(np.float32)
        with GridSurrogate(model, ny, nx, precision="bf16") as sur:
            f = sur.solve(grid)[0]
        emu = orc.solve_grid(grid.astype(np.float64), oracle_model(model), precision="bf16")
        assert rel_l2(f, emu.fields) <= 5e-3


@pytest.mark.parametrize("variant,p_in,p_out,arch_layers", [("deltas", 1, 1, None), ("gradp", 3, 130, None), ("chapter5", 160, 2, None),
                                                            ("deltas", 40, 24, 1), ("deltas", 16, 16, 20)])
def test_extreme_component_and_layer_counts(variant, p_in, p_out, arch_layers):
    """One retained component, more than 128 output components (generic decode path), more than 128 input components
    (general encode path), a single Dense layer (head only) and a 20-layer stack (MLP_huger depth)."""
    model = synthetic.make_model(variant, p_in=p_in, p_out=p_out, seed_pca=300 + p_in, seed_w=p_out)
    if arch_layers is not None:
        widths = [] if arch_layers == 1 else [64] * (arch_layers - 1)
        model.weights = synthetic.he_dense_stack(p_in, widths, p_out, 9)
    grid = synthetic.channel_grid(256, 256, seed=30).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        fields = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.abs(fields - sol.fields).max() <= 1e-4 * max(np.abs(sol.fields).max(), 1e-3)


def test_cli_default_cap_of_512_components():
    """`max_number_PC = 512` is the cap the reference's evaluator mains pass by default (Eval_dual_Dense_onlycil.py main,
    entry_point.py --max_num_PC): 512 components on both sides -- 16 component tiles in the encode, the generic decode, a
    first Dense layer of 512 inputs -- general and geometry-bound path."""
    model = synthetic.make_model("deltas", p_in=512, p_out=512, seed_pca=512, seed_w=5)
    grid = synthetic.channel_grid(256, 256, seed=31).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        fields = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
        assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        if sur.bind_geometry(grid):                          # binding needs <= 128 output components: not for this model
            assert np.abs(sur.solve(grid)[0] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    model = synthetic.make_model("gradp", p_in=512, p_out=96, seed_pca=513, seed_w=6)
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(grid)
        bound = sur.solve(grid)[0]
    sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
    assert np.abs(bound - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_maximum_component_counts_of_the_abi():
    """psm_create accepts up to 1024 components per side: 1024 in (32 component tiles, the separate slab-reduce launch, a first
    Dense layer of 1024 inputs) and 640 out (generic decode, 20 groups of 32) on the Chapter-5 layout."""
    model = synthetic.make_model("chapter5", p_in=1024, p_out=640, seed_pca=1024, seed_w=7)
    grid = synthetic.channel_grid(256, 300, seed=32).astype(np.float32)
    with GridSurrogate(model, 256, 300) as sur:
        fields = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    with pytest.raises(_lib.PsmError):
        GridSurrogate(synthetic.make_model("deltas", p_in=1025, p_out=8, seed_pca=3), 256, 256)


def test_all_solid_and_all_flow_grids():
    """No flow cell at all: every masked strip is empty, the reference's np.mean([]) = NaN propagates through the
    offsets to the whole field (NumPy semantics, reproduced); no solid cell at all: nothing is masked."""
    model = synthetic.make_model("deltas", p_in=8, p_out=8)
    om = oracle_model(model)
    solid = synthetic.channel_grid(256, 256, seed=3).astype(np.float32)
    solid[..., 2] = 0.0
    flow = synthetic.channel_grid(256, 256, seed=3, obstacle="none").astype(np.float32)
    flow[..., 2] = np.maximum(flow[..., 2], 1e-3)
    with GridSurrogate(model, 256, 256) as sur:
        for g in (solid, flow):
            got = sur.solve(g)[0]
            with np.errstate(all="ignore"):
                ref = orc.solve_grid(g.astype(np.float64), om).fields
            np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
            ok = ~np.isnan(ref)
            if ok.any():
                assert np.abs(got[ok] - ref[ok]).max() <= 1e-4 * np.abs(ref[ok]).max()
        assert np.isnan(sur.solve(solid)[0]).any()


def test_gradp_first_block_without_a_flow_cell_is_reference_undefined():
    """A solid region covering the whole first block of a U_to_gradP grid: the reference's search for the first column holding
    a flow cell runs off the block and fails its own assert (UGP:294-300, "At least the right-most column ... must belong to the
    flow domain, or this won't work"); the oracle raises.  The library neither fails nor returns a plausible dp/dx: the
    reference BC is NaN (psm_plan.h, first_col_mean), so every dp/dx value is NaN, while dp/dy -- anchored on row 1 of the
    block (UGP:302-303), which np.mean([]) turns into NaN as well when the block is solid -- follows NumPy semantics.  Found by
    tests/measure/soak.py (seed 9041)."""
    model = synthetic.make_model("gradp", p_in=12, p_out=10)
    om = oracle_model(model)
    g = synthetic.channel_grid(272, 288, seed=5, obstacle="none").astype(np.float32)
    g[:128, :128, :] = 0.0
    with pytest.raises(ValueError, match="first block has no flow cell"):
        orc.solve_grid(g.astype(np.float64), om)
    flow = g[..., model.sdf_ch] != 0
    with GridSurrogate(model, 272, 288) as sur:
        general = sur.solve(g)[0]
        assert np.isnan(general[..., 0][flow]).all()
        assert sur.bind_geometry(g)
        bound = sur.solve(g)[0]
        np.testing.assert_array_equal(np.isnan(bound), np.isnan(general))
        ok = ~np.isnan(general)
        if ok.any():
            assert np.abs(bound[ok] - general[ok]).max() <= 2e-5 * np.abs(general[ok]).max()
        assert sur.guard_trips == 0


def test_chapter4_channel_configuration():
    """The Chapter-4 M_fU evaluator's shape (Thesis_Work/Chapter4/MLP/M_fU/Evaluation/Eval.py:205-223): two input channels
    (f(U), SDF) with the flow mask in channel 1, Chapter-5 block layout, 116 -> 39 components like its model_first_.h5."""
    model = synthetic.make_model("chapter5", p_in=116, p_out=39, c_in=2, seed_pca=404, seed_w=4)
    model.sdf_ch = 1
    g3 = synthetic.channel_grid(300, 400, seed=14)
    grid = np.stack([g3[..., 0] * 0.7 + g3[..., 1] * 0.3, g3[..., 2]], axis=-1).astype(np.float32)
    with GridSurrogate(model, 300, 400) as sur:
        fields = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
    assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_reassembly_on_random_shapes_and_obstacles():
    """a12 (offset chain + paste + global shift) on 24 seeded random grid shapes per variant mix, with random solid
    bands that empty some overlap strips (the np.isnan branches) -- device result against the oracle's serial
    restatement, NaN pattern included."""
    rng = np.random.default_rng(2024)
    variants = ("deltas", "gradp", "chapter5")
    for trial in range(24):
        variant = variants[trial % 3]
        ny, nx = int(rng.integers(130, 520)), int(rng.integers(260, 900))
        try:
            lay = orc.block_layout(variant, ny, nx)
        except Exception:
            continue
        model = synthetic.make_model(variant, p_in=4, p_out=4)
        grid = synthetic.channel_grid(ny, nx, seed=100 + trial, obstacle=("circle", "rectangle", "plate", "none")[trial % 4],
                                      cx=float(rng.uniform(0.2, 0.8)), cy=float(rng.uniform(0.2, 0.8))).astype(np.float32)
        for _ in range(int(rng.integers(0, 3))):                       # solid bands: whole strips without a flow cell
            y0, x0 = int(rng.integers(0, ny - 40)), int(rng.integers(0, nx - 140))
            grid[y0:y0 + int(rng.integers(8, 40)), x0:x0 + int(rng.integers(100, 140)), :] = 0.0
        c_out = model.c_out
        bp = rng.standard_normal((lay.B, 128, 128, c_out)).astype(np.float32)
        xb = orc.extract_blocks(grid.astype(np.float64), lay, 3)
        try:
            sur = GridSurrogate(model, ny, nx)
        except _lib.PsmError:
            continue                                                   # shapes the reference itself cannot process
        with sur:
            f = sur.reassemble(grid, bp)
        for c in range(c_out):
            with np.errstate(all="ignore"):
                if variant == "gradp":
                    ref = orc.assemble_gradp(("dp_dx", "dp_dy")[c], bp[..., c], xb, lay, degenerate="skip").field
                elif variant == "deltas":
                    ref = orc.assemble_deltas(bp[..., c], xb, lay, degenerate="skip").field
                else:
                    ref = orc.assemble_chapter5(bp[..., c], xb, lay).field
            assert np.array_equal(np.isnan(f[..., c]), np.isnan(ref)), (variant, ny, nx, trial)
            ok = ~np.isnan(ref)
            if ok.any():
                assert np.abs(f[..., c][ok] - ref[ok]).max() <= 3e-4 * max(1.0, np.abs(ref[ok]).max()), (variant, ny, nx, trial)


@pytest.mark.parametrize("p_in", [45, 20, 70])
def test_streamed_encode_with_few_component_tiles(p_in):
    """Many block rows (> 32: the streamed 64-row-chunk form of the encode kernel) with 2, 1 and 3 tiles of 32 input
    components -- the reference's own network has 45: the waves split the row tiles between them; coefficients and
    fields against the oracle, case by case."""
    model = synthetic.make_model("deltas", p_in=p_in, p_out=32)
    grids = synthetic.random_obstacle_cases(5, 256, 256, seed=13).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=5) as sur:
        fields = sur.solve(grids)
        for k in (0, 4):
            sol = orc.solve_grid(grids[k].astype(np.float64), oracle_model(model))
            check_against_oracle(sur, grids[k], model, sol, n_cases=5, case=k)
            assert np.abs(fields[k] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
