"""Geometry-bound solves (psm_bind_geometry: strip means from dot products with the last hidden activation, decode +
offset chain + paste in one launch) against the general 8-launch path, the oracle and the golden vectors produced by the
reference's own statements.  Needs a real MI355X.

Tolerances: the bound path sums the same float32 products in another order (the strip means go through tables folded
with the head layer), so it is compared with the general path at 2e-5 * max|field| and with the oracle / golden
fields at the tolerances of tests/test_gpu_parity.py."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate, synthetic
from hipmem import DeviceArray
from test_gpu_parity import oracle_model, rel_l2

pytestmark = pytest.mark.gpu


def same(a, b, tol=2e-5):
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    scale = max(np.nanmax(np.abs(b)), 1e-6) if np.isfinite(b).any() else 1.0
    assert np.nanmax(np.abs(a - b), initial=0.0) <= tol * scale, (np.nanmax(np.abs(a - b)), scale)


@pytest.mark.parametrize("name", cases.DENSE_GOLDEN_CASES)
def test_bound_equals_general_path_and_golden(name):
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    out_scale = [model.out_scale] if model.variant == "deltas" else None
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, grid.shape[0], grid.shape[1]) as sur:
        general = sur.solve(g32, out_scale=out_scale)[0]
        offs_general = sur.stage("offsets")[0].copy()
        assert sur.bind_geometry(g32) and sur.geometry_bound
        bound = sur.solve(g32, out_scale=out_scale)[0]
        offs_bound = sur.stage("offsets")[0]
        same(offs_bound, offs_general)
        same(bound, general)
        # other velocities on the same geometry: only the SDF channel is part of the binding
        g2 = g32.copy()
        rng = np.random.default_rng(5)
        for ch in range(g2.shape[-1]):
            if ch != model.sdf_ch:
                g2[..., ch] = (g2[..., ch] * 0.7 + 0.05 * rng.standard_normal(g2.shape[:2])).astype(np.float32) * (g2[..., model.sdf_ch] != 0)
        b2 = sur.solve(g2, out_scale=out_scale)[0]
        sur.unbind_geometry()
        assert not sur.geometry_bound
        same(b2, sur.solve(g2, out_scale=out_scale)[0])
    ref = gold["fields"]
    assert np.abs(bound - ref).max() <= 2e-4 * np.abs(ref).max()
    assert rel_l2(bound, ref) <= 5e-5


def test_config1_bound_against_oracle():
    """BASELINE config 1 on the bound path: 256x256 U_to_gradP, P_i = P_o = 128, MLP_small."""
    model = synthetic.make_model("gradp")
    grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(grid)
        fields = sur.solve(grid)[0]
        offs = sur.stage("offsets")[0]
    for c, a in enumerate(sol.assemblies):
        np.testing.assert_allclose(offs[c], a.offsets, rtol=0, atol=1e-4 * np.nanmax(np.abs(a.field)), equal_nan=True)
    assert np.isfinite(fields).all()
    assert np.abs(fields - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()


def test_config2_sequence_on_a_bound_geometry():
    """BASELINE config 2: sequential deltaU_to_deltaP solves on one geometry with a different out_scale per step."""
    model = synthetic.make_model("deltas")
    base = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur, GridSurrogate(model, 256, 256) as ref:
        assert sur.bind_geometry(base)
        for step in range(4):
            g = synthetic.delta_grid(256, 256, seed=2, step=step).astype(np.float32)
            g[..., model.sdf_ch] = base[..., model.sdf_ch]
            g[..., :model.sdf_ch] *= (base[..., model.sdf_ch:model.sdf_ch + 1] != 0)
            sc = [0.5 + 0.25 * step]
            same(sur.solve(g, out_scale=sc)[0], ref.solve(g, out_scale=sc)[0])


def test_all_solid_and_all_flow_grids_bound():
    """Empty masks everywhere (every strip mean NaN) and full masks: the NaN logic of the chain only depends on the
    counts, which the binding tabulates."""
    for variant in ("deltas", "gradp", "chapter5"):
        model = synthetic.make_model(variant, p_in=16, p_out=16)
        for fill in (0.0, 1.0):
            g = synthetic.channel_grid(256, 256, seed=3).astype(np.float32)
            g[..., model.sdf_ch] = fill
            if fill == 0.0:
                g[...] = 0.0
            with GridSurrogate(model, 256, 256) as sur:
                general = sur.solve(g)[0]
                assert sur.bind_geometry(g)
                same(sur.solve(g)[0], general)


def test_bind_lifecycle_and_unsupported_configurations():
    model = synthetic.make_model("deltas", p_in=16, p_out=16)
    g = synthetic.channel_grid(256, 256, seed=4).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=2) as sur:
        assert sur.bind_geometry(g)
        one = sur.solve(g)[0]
        two = sur.solve(np.stack([g, g]))           # case batches keep the general path
        same(two[0], one); same(two[1], one)
        with pytest.raises(ValueError):
            sur.bind_geometry(g[:100])
        # opt-in contract check of the host-grid entries: a grid of another geometry drops the binding
        other = synthetic.channel_grid(256, 256, seed=8, cx=0.65).astype(np.float32)
        assert sur.bind_geometry(g)
        sur.check_bound = True
        got = sur.solve(other)[0]
        assert not sur.geometry_bound
        sur.check_bound = False
        same(got, sur.solve(other)[0])
        # the same check after a bind from DEVICE memory: the library hands back the pattern it bound (psm_bound_mask)
        d_g = DeviceArray(g)
        assert sur.bind_geometry(d_g.ptr, on_device=True)
        np.testing.assert_array_equal(sur._bound_mask[0], g[..., 2] != 0)
        sur.check_bound = True
        sur.solve(g)
        assert sur.geometry_bound                       # same geometry: the binding stays
        sur.solve(other)
        assert not sur.geometry_bound
        sur.check_bound = False
        # a new plan or a model change drops the binding (its tables belong to the old block layout / head layer)
        assert sur.bind_geometry(g) and sur.geometry_bound
        sur._chk(sur.lib.psm_plan_grid(sur.h, 256, 256))
        assert not sur.geometry_bound
        same(sur.solve(g)[0], one)
        assert sur.bind_geometry(g)
        W, b = model.weights[-1]
        import ctypes as C
        W2 = np.ascontiguousarray(W * 0.5, np.float32); b2 = np.ascontiguousarray(b, np.float32)
        sur._chk(sur.lib.psm_set_dense(sur.h, len(model.weights) - 1, W2.shape[0], W2.shape[1],
                                       W2.ctypes.data_as(C.POINTER(C.c_float)), b2.ctypes.data_as(C.POINTER(C.c_float))))
        assert not sur.geometry_bound
    big = synthetic.make_model("deltas", p_in=16, p_out=16)
    gb = synthetic.channel_grid(128, 128 + 96 * 70, seed=5).astype(np.float32)      # > 64 block columns
    with GridSurrogate(big, gb.shape[0], gb.shape[1]) as sur:
        assert sur.bind_geometry(gb) is False and not sur.geometry_bound
        assert np.isfinite(sur.solve(gb)[0]).all()
    with GridSurrogate(model, 256, 256, precision="bf16", max_cases=2) as sur:
        assert sur.bind_geometry(np.stack([g, g])) is True
        assert sur.bind_geometry(g) is True
    wide = synthetic.make_model("gradp", p_in=16, p_out=130)                         # > 128 output components
    with GridSurrogate(wide, 256, 256) as sur:
        assert sur.bind_geometry(g) is False


def test_bound_unaligned_grid_and_device_pointer():
    from hipmem import DeviceArray
    model = synthetic.make_model("chapter5", p_in=24, p_out=24)
    g = synthetic.channel_grid(300, 402, seed=6).astype(np.float32)
    with GridSurrogate(model, 300, 402) as sur:
        general = sur.solve(g)[0]
        d_in, d_out = DeviceArray(g[None]), DeviceArray(shape=(1, 300, 402, 1))
        assert sur.bind_geometry(d_in.ptr, on_device=True)
        sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        same(d_out.numpy()[0], general)


def test_bound_path_on_random_shapes_and_obstacles():
    """Full solves on 24 seeded random grid shapes (all three variants) with random obstacles and solid bands that
    empty some overlap strips (the np.isnan branches of the chain, p_i == 0 duplicate rows, unaligned widths):
    geometry-bound result against the general path, NaN pattern included."""
    from psm_amd import _lib
    rng = np.random.default_rng(77)
    variants = ("deltas", "gradp", "chapter5")
    n_bound = 0
    for trial in range(24):
        variant = variants[trial % 3]
        ny, nx = int(rng.integers(130, 520)), int(rng.integers(260, 900))
        model = synthetic.make_model(variant, p_in=int(rng.integers(3, 40)), p_out=int(rng.integers(3, 130)))
        grid = synthetic.channel_grid(ny, nx, seed=300 + trial, obstacle=("circle", "rectangle", "plate", "none")[trial % 4],
                                      cx=float(rng.uniform(0.2, 0.8)), cy=float(rng.uniform(0.2, 0.8))).astype(np.float32)
        for _ in range(int(rng.integers(0, 3))):
            y0, x0 = int(rng.integers(0, ny - 40)), int(rng.integers(0, nx - 140))
            grid[y0:y0 + int(rng.integers(8, 40)), x0:x0 + int(rng.integers(100, 140)), :] = 0.0
        try:
            sur = GridSurrogate(model, ny, nx)
        except _lib.PsmError:
            continue                                                   # shapes the reference itself cannot process
        with sur:
            sc = [float(rng.uniform(0.3, 2.0))]
            general = sur.solve(grid, out_scale=sc)[0]
            if not sur.bind_geometry(grid):
                continue
            n_bound += 1
            bound = sur.solve(grid, out_scale=sc)[0]
            offs_b = sur.stage("offsets")[0]
        assert np.array_equal(np.isnan(bound), np.isnan(general)), (variant, ny, nx, trial)
        ok = ~np.isnan(general)
        if ok.any():
            assert np.abs(bound[ok] - general[ok]).max() <= 5e-5 * max(1e-3, np.abs(general[ok]).max()), (variant, ny, nx, trial)
    assert n_bound >= 15


@pytest.mark.parametrize("variant,n", [("deltas", 8), ("gradp", 3), ("chapter5", 5), ("deltas", 40)])
def test_bound_case_batch_equals_general_path(variant, n):
    """BASELINE config 3 shape: a batch of random-obstacle cases, one geometry per case slot (7 launches instead of 9)."""
    model = synthetic.make_model(variant, p_in=32, p_out=32)
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=3).astype(np.float32)
    sc = list(np.linspace(0.5, 1.5, n).astype(np.float32))
    with GridSurrogate(model, 256, 256, max_cases=n) as sur:
        general = sur.solve(grids, out_scale=sc)
        assert sur.bind_geometry(grids)
        bound = sur.solve(grids, out_scale=sc)
        for k in range(n):
            same(bound[k], general[k])
        # fewer cases than bound: the general path, still right
        same(sur.solve(grids[:2], out_scale=sc[:2])[1], general[1])
        # velocities change, geometries stay
        g2 = grids.copy()
        g2[..., :model.sdf_ch] *= 0.5
        b2 = sur.solve(g2, out_scale=sc)
        sur.unbind_geometry()
        r2 = sur.solve(g2, out_scale=sc)
        for k in range(n):
            same(b2[k], r2[k])


def test_bound_device_api_graph_profile_and_timing_agree(monkeypatch):
    """Stream launches, hipGraph replay (PSM_GRAPH=1), the event-profiled pass and the encode-timing pass all run the
    bound sequence and give bit-identical fields; re-binding another geometry replaces the captured graph."""
    from hipmem import DeviceArray
    model = synthetic.make_model("gradp", p_in=64, p_out=64)
    grid = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    other = synthetic.channel_grid(256, 256, seed=2, cx=0.6).astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(grid)
        host = sur.solve(grid)[0]
        d_in, d_out = DeviceArray(grid), DeviceArray(shape=(256, 256, 2), dtype=np.float32)
        for _ in range(3):
            sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        np.testing.assert_array_equal(d_out.numpy(), host)
        ms = sur.profile(d_in.ptr, 1, d_out.ptr)
        np.testing.assert_array_equal(d_out.numpy(), host)
        assert ms["encode"] > 0 and ms["decode"] > 0
        sur.enable_kernel_timing("encode")
        sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        total, n = sur.kernel_timing("encode")
        sur.enable_kernel_timing("encode", False)
        assert n == 1 and total > 0
        np.testing.assert_array_equal(d_out.numpy(), host)
        sur.unbind_geometry()
        general_other = sur.solve(other)[0]
    monkeypatch.setenv("PSM_GRAPH", "1")
    with GridSurrogate(model, 256, 256) as sur:
        assert sur.bind_geometry(grid)
        for _ in range(3):
            sur.solve_device(d_in.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        np.testing.assert_array_equal(d_out.numpy(), host)
        d_in2 = DeviceArray(other)
        assert sur.bind_geometry(other)                 # new tables: the captured graph must not be replayed
        for _ in range(2):
            sur.solve_device(d_in2.ptr, 1, d_out.ptr, 0)
        sur.synchronize()
        same(d_out.numpy(), general_other)
        d_in2.free()
    d_in.free(); d_out.free()


@pytest.mark.parametrize("variant,ny,nx", [("deltas", 512, 512), ("gradp", 256, 256), ("chapter5", 300, 400)])
def test_bf16_bound_equals_bf16_general_path(variant, ny, nx):
    """BASELINE config 4 precision on a bound geometry (7 launches: the strip dots come from the bf16-rounded res in a
    launch of their own, so the rounding points are those of the bf16 decode): same fields as the general bf16 path
    up to float32 summation order, and within the bf16 tolerance of the float64 oracle."""
    model = synthetic.make_model(variant, p_in=48, p_out=40)
    grid = synthetic.channel_grid(ny, nx, seed=4, noise=0.05).astype(np.float32)
    with GridSurrogate(model, ny, nx, precision="bf16") as sur:
        general = sur.solve(grid)[0]
        assert sur.bind_geometry(grid)
        bound = sur.solve(grid)[0]
        g2 = grid.copy(); g2[..., :model.sdf_ch] *= 0.8
        b2 = sur.solve(g2)[0]
        sur.unbind_geometry()
        r2 = sur.solve(g2)[0]
    same(bound, general, tol=5e-5)
    same(b2, r2, tol=5e-5)
    sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
    assert rel_l2(bound, sol.fields) <= 2e-2


def test_bound_four_channels_narrow_hidden_layer_and_min_max_scaler():
    """pressureSM_Poisson shape (C_in = 4, SDF in channel 3, min_max scaler) and an architecture whose last hidden layer
    is not 512 wide (MLP_small_unet ends 256 -> 512; MLP_big ends ... -> 256): the folded tables follow the head's width."""
    m4 = synthetic.make_model("deltas", p_in=48, p_out=32, c_in=4, scaler_kind="min_max")
    m4.sdf_ch = 3
    g3 = synthetic.channel_grid(256, 320, seed=5)
    g4 = np.concatenate([g3[..., :1] * g3[..., 1:2], g3], axis=-1).astype(np.float32)
    with GridSurrogate(m4, 256, 320) as sur:
        general = sur.solve(g4)[0]
        assert sur.bind_geometry(g4)
        same(sur.solve(g4)[0], general)
    sol = orc.solve_grid(g4.astype(np.float64), oracle_model(m4))
    assert np.abs(general - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
    for arch in ("MLP_big", "MLP_small_unet"):
        model = synthetic.make_model("gradp", p_in=20, p_out=24, arch=arch)
        grid = synthetic.channel_grid(256, 256, seed=6).astype(np.float32)
        with GridSurrogate(model, 256, 256) as sur:
            general = sur.solve(grid)[0]
            assert sur.bind_geometry(grid)
            same(sur.solve(grid)[0], general)


@pytest.mark.parametrize("variant,ny,nx", [("chapter5", 400, 3000), ("gradp", 512, 512), ("deltas", 300, 2100)])
def test_bound_with_more_than_64_blocks(variant, ny, nx):
    """The reference's shipped case is a 400 x 3000 grid (104 blocks in the Chapter-5 layout, python_module.py:306-329);
    U_to_gradP at 512 x 512 has 182.  More than 64 blocks take the two-launch bound form (chain launch + chunked
    decode + paste): 7 launches instead of 9."""
    model = synthetic.make_model(variant, p_in=45, p_out=48)
    grid = synthetic.channel_grid(ny, nx, seed=12).astype(np.float32)
    with GridSurrogate(model, ny, nx) as sur:
        assert sur.B > 64
        general = sur.solve(grid)[0]
        offs_general = sur.stage("offsets")[0].copy()
        assert sur.bind_geometry(grid)
        bound = sur.solve(grid)[0]
        same(sur.stage("offsets")[0], offs_general)
        same(bound, general)
        g2 = grid.copy(); g2[..., :model.sdf_ch] *= 0.6
        b2 = sur.solve(g2)[0]
        sur.unbind_geometry()
        same(b2, sur.solve(g2)[0])


@pytest.mark.parametrize("variant,ny,nx,n", [("deltas", 256, 256, 8), ("gradp", 512, 512, 1), ("chapter5", 400, 1500, 2)])
def test_bf16_bound_batches_and_many_blocks(variant, ny, nx, n):
    """bf16 handles on the two-launch bound form (case batches, more than 64 blocks): the general bf16 path's fields."""
    model = synthetic.make_model(variant, p_in=40, p_out=40)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=30 + k, cx=0.3 + 0.05 * k).astype(np.float32) for k in range(n)])
    with GridSurrogate(model, ny, nx, max_cases=n, precision="bf16") as sur:
        general = sur.solve(grids)
        assert sur.bind_geometry(grids)
        bound = sur.solve(grids)
    for k in range(n):
        same(bound[k], general[k], tol=5e-5)


@pytest.mark.parametrize("mode", ["0", "3"])
def test_x6_arithmetic_against_the_oracle(mode, monkeypatch):
    """The PCA contractions on the bf16 matrix pipe at float32 accuracy (x6: every float32 operand split exactly into three
    bf16 terms, six MFMA terms kept; default for the encode from two row tiles up and for the bound decode) against the
    exact-float32 MFMA (PSM_X6=0) and the float64 oracle: same tolerances, batch of 8 cases and single case, both forced
    everywhere (PSM_X6=3) and switched off."""
    monkeypatch.setenv("PSM_X6", mode)
    from test_gpu_parity import check_against_oracle
    model = synthetic.make_model("deltas")
    grids = synthetic.random_obstacle_cases(8, 256, 256, seed=3).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=8) as sur:
        general = sur.solve(grids)
        sol = orc.solve_grid(grids[5].astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grids[5], model, sol, n_cases=8, case=5)            # coefficients <= 2e-6, decoded blocks <= 1e-5
        assert sur.bind_geometry(grids)
        bound = sur.solve(grids)
        assert np.abs(bound[5] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        same(bound, general)
    m1 = synthetic.make_model("gradp")
    g1 = synthetic.channel_grid(256, 256, seed=1).astype(np.float32)
    with GridSurrogate(m1, 256, 256) as sur:
        f = sur.solve(g1)[0]
        s1 = orc.solve_grid(g1.astype(np.float64), oracle_model(m1))
        check_against_oracle(sur, g1, m1, s1)
        assert sur.bind_geometry(g1)
        fb = sur.solve(g1)[0]
    assert rel_l2(fb, s1.fields) <= 2e-6 and rel_l2(f, s1.fields) <= 2e-6
