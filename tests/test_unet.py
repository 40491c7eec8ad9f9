"""Convolutional surrogate path (SURVEY.md §8 row a-conv; parity UNPINNED: the reference has no such network).
CPU: the NumPy oracle of the build-defined UNet-S against torch's conv2d / max_pool2d / interpolate.
GPU: the HIP kernels (psm_unet_*, through the C-ABI) against the oracle, layer by layer and end to end.

Tolerance: the device computes exact float32 products with float32 accumulation (v_mfma_f32_16x16x4_f32), the
oracle accumulates in float64 and rounds every activation to float32: per-layer max-abs <= 2e-5 * max|activation|,
final field <= 1e-4 * max|field| (19 layers deep)."""
import numpy as np
import pytest

from oracle import unet_oracle as uo
from psm_amd import synthetic


def _torch_forward(g, W):
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(g).permute(2, 0, 1)[None]

    def conv(x, i, relu):
        w, b = W[i]
        y = F.conv2d(x, torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), padding=w.shape[0] // 2)
        return F.relu(y) if relu else y
    enc, i = [], 0
    for l in range(5):
        if l > 0:
            x = F.max_pool2d(x, 2)
        x = conv(x, i, True); x = conv(x, i + 1, True); i += 2
        enc.append(x)
    for l in range(3, -1, -1):
        x = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), enc[l]], 1)
        x = conv(x, i, True); x = conv(x, i + 1, True); i += 2
    return conv(x, i, False)[0].permute(1, 2, 0).numpy()


def test_product_weight_generator_matches_the_oracle_spec():
    specs = uo.unet_specs(4, (16, 32, 48), 2)
    assert synthetic.unet_conv_shapes(4, (16, 32, 48), 2) == [(s.k, s.c_in, s.c_out) for s in specs]
    for (a, b), (c, d) in zip(synthetic.unet_he_weights(4, (16, 32, 48), 2, seed=5), uo.he_weights(specs, seed=5)):
        np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)


def test_spec_and_flops():
    specs = uo.unet_specs()
    assert len(specs) == 19 and [s.name for s in specs][:4] == ["enc0a", "enc0b", "enc1a", "enc1b"]
    assert specs[10].name == "dec3a" and specs[10].c_in == 256 + 128 and specs[-1].k == 1
    assert abs(uo.unet_flops(256, 256) / 1e9 - 7.0) < 0.01                  # 7.0 GFLOP per solve


def test_oracle_matches_torch_cpu():
    W = uo.he_weights(uo.unet_specs(), seed=7)
    g = synthetic.channel_grid(64, 96, seed=3).astype(np.float32)
    y = uo.unet_forward(g, W)
    t = _torch_forward(g, W)
    assert y.shape == (64, 96, 1)
    assert np.abs(y - t).max() <= 2e-5 * np.abs(t).max()


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx,n", [(256, 256, 1), (96, 160, 2), (16, 48, 1)])
def test_gpu_unet_matches_oracle_layer_by_layer(ny, nx, n):
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=11)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=20 + k).astype(np.float32) for k in range(n)])
    with UNetSurrogate(W, ny, nx, max_cases=n) as net:
        assert net.flops == uo.unet_flops(ny, nx)
        out = net.forward(grids)
        assert out.shape == (n, ny, nx, 1)
        for k in range(n):
            ref, acts = uo.unet_forward(grids[k], W, return_all=True)
            for i in range(len(specs) - 1):
                a = net.activation(i, n)[k]
                assert a.shape == acts[i].shape, specs[i].name
                assert np.abs(a - acts[i]).max() <= 2e-5 * max(np.abs(acts[i]).max(), 1e-6), specs[i].name
            assert np.abs(out[k] - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.gpu
def test_gpu_unet_other_widths_and_errors():
    from psm_amd import UNetSurrogate, _lib
    widths = (32, 48, 80)                                   # three levels, channel counts that are not powers of two
    specs = uo.unet_specs(4, widths, 2)
    W = uo.he_weights(specs, seed=5)
    g = np.random.default_rng(0).standard_normal((40, 72, 4)).astype(np.float32)
    with UNetSurrogate(W, 40, 72, c_in=4, c_out=2, widths=widths) as net:
        out = net.forward(g)[0]
    ref = uo.unet_forward(g, W, widths)
    assert np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()
    with pytest.raises(_lib.PsmError):
        UNetSurrogate(W, 42, 72, c_in=4, c_out=2, widths=widths)           # 42 is not a multiple of 4
    with pytest.raises(ValueError):
        UNetSurrogate(W[:-1], 40, 72, c_in=4, c_out=2, widths=widths)


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx,n", [(256, 256, 1), (96, 160, 2)])
def test_gpu_unet_bf16_matches_bf16_oracle(ny, nx, n):
    """bf16 operand path (v_mfma_f32_16x16x32_bf16): against the oracle with the same rounding points (inputs and
    kernels of every 3x3 convolution to bf16, wide accumulation).  An activation that sits on a bf16 rounding
    boundary can round the other way after a 1e-7 difference upstream, so the bound is a few bf16 ulps of the
    layer's range, not float32 epsilon; the float32 oracle is only a sanity bound."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=11)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=20 + k).astype(np.float32) for k in range(n)])
    with UNetSurrogate(W, ny, nx, max_cases=n, precision="bf16", keep_activations=True) as net:
        out = net.forward(grids)
        for k in range(n):
            ref, acts = uo.unet_forward(grids[k], W, return_all=True, precision="bf16")
            for i in range(len(specs) - 1):
                a = net.activation(i, n)[k]
                err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
                assert err <= 1e-2, (specs[i].name, err)            # bf16 epsilon is 7.8e-3
            assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 1e-2
            f32 = uo.unet_forward(grids[k], W)
            assert np.linalg.norm(out[k] - f32) / np.linalg.norm(f32) <= 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx,n,keep", [(96, 160, 2, True), (64, 80, 3, True), (112, 240, 1, False), (256, 256, 2, False)])
def test_gpu_unet_bf16_fused_level_pairs(ny, nx, n, keep, monkeypatch):
    """bf16 mode, the two convolutions of a level in one launch (psm_unet_pair.hip; forced for every eligible level with
    PSM_UNET_PAIR_MIN=1): image sizes that are not multiples of the 30 x 14 tile, several cases.  Same rounding points
    as the unfused path, so the same bounds against the bf16 oracle.  Without keep_activations the inner activations are
    never stored and asking for them is an error."""
    from psm_amd import UNetSurrogate, _lib
    monkeypatch.setenv("PSM_UNET_PAIR_MIN", "1")
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=13)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=40 + k).astype(np.float32) for k in range(n)])
    with UNetSurrogate(W, ny, nx, max_cases=n, precision="bf16", keep_activations=keep) as net:
        out = net.forward(grids)
        for k in range(n):
            ref, acts = uo.unet_forward(grids[k], W, return_all=True, precision="bf16")
            on_chip = []
            for i in range(len(specs) - 1):
                try:
                    a = net.activation(i, n)[k]
                except _lib.PsmError:                              # kept on chip by a fused pair (never with keep_activations)
                    assert not keep
                    on_chip.append(i)
                    continue
                err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
                assert err <= 1e-2, (specs[i].name, err)
            if not keep:                                           # the level-0 pairs at least: enc0a, dec0a and dec0b (head fused); the
                assert {0, 16, 17} <= set(on_chip) <= {0, 2, 14, 16, 17}, on_chip     # 128^2 pairs depend on what the planner splits
            assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 1e-2
    monkeypatch.setenv("PSM_UNET_NO_PAIR", "1")                    # and the unfused path gives the same field to bf16 rounding flips
    with UNetSurrogate(W, ny, nx, max_cases=n, precision="bf16") as net:
        out2 = net.forward(grids)
    assert np.linalg.norm(out - out2) / np.linalg.norm(out2) <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("pairs", ["planner", "forced", "off"])
def test_gpu_unet_bf16_512x512_layer_by_layer(pairs, monkeypatch):
    """BASELINE configs[4] as BASELINE.json words it -- 512x512, bf16 MFMA conv path, batch 1: every layer against the
    bf16-rounding oracle, with the pair kernels as the planner places them, forced for every eligible level, and off."""
    from psm_amd import UNetSurrogate
    if pairs == "forced":
        monkeypatch.setenv("PSM_UNET_PAIR_MIN", "1")
    elif pairs == "off":
        monkeypatch.setenv("PSM_UNET_NO_PAIR", "1")
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=11)
    g = synthetic.channel_grid(512, 512, seed=4, noise=0.05).astype(np.float32)
    ref, acts = uo.unet_forward(g, W, return_all=True, precision="bf16")
    with UNetSurrogate(W, 512, 512, precision="bf16", keep_activations=True) as net:
        out = net.forward(g)[0]
        for i in range(len(specs) - 1):
            a = net.activation(i, 1)[0]
            assert a.shape == acts[i].shape, specs[i].name
            err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
            assert err <= 1e-2, (specs[i].name, err)
    assert np.isfinite(out).all()
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= 1e-2
    with UNetSurrogate(W, 512, 512, precision="bf16") as net:       # the timed configuration: inner activations on chip
        out2 = net.forward(g)[0]
    assert np.linalg.norm(out2 - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.gpu
def test_gpu_unet_bf16_pairs_512x512_eight_cases():
    """Pair kernels at 512x512 x 8 cases per step (the largest launch the bench times: 2312 tiles per level-0 pair): first,
    middle and last case against the bf16 oracle end to end, the last one layer by layer as well."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=11)
    grids = np.stack([synthetic.channel_grid(512, 512, seed=60 + k, noise=0.05, cx=0.25 + 0.06 * k).astype(np.float32) for k in range(8)])
    with UNetSurrogate(W, 512, 512, max_cases=8, precision="bf16", keep_activations=True) as net:
        out = net.forward(grids)
        ref, acts = uo.unet_forward(grids[7], W, return_all=True, precision="bf16")
        for i in range(len(specs) - 1):
            a = net.activation(i, 8)[7]
            err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
            assert err <= 1e-2, (specs[i].name, err)
        assert np.linalg.norm(out[7] - ref) / np.linalg.norm(ref) <= 1e-2
    with UNetSurrogate(W, 512, 512, max_cases=8, precision="bf16") as net:
        out2 = net.forward(grids)
    assert np.isfinite(out2).all()
    for k in (0, 4, 7):
        ref = uo.unet_forward(grids[k], W, precision="bf16")
        assert np.linalg.norm(out2[k] - ref) / np.linalg.norm(ref) <= 1e-2, k
    assert np.linalg.norm(out2 - out) / np.linalg.norm(out) <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nx", [(144, 208), (176, 144)])
def test_gpu_unet_bf16_pairs_tiles_straddle_the_image_edge_on_both_axes(ny, nx, monkeypatch):
    """The pair kernels own 30 x 14 output tiles: image sizes that leave a partial tile at the bottom AND at the right at
    both fused levels (144 = 4*30 + 24, 208 = 14*14 + 12; 72 = 2*30 + 12, 104 = 7*14 + 6; 176 x 144 likewise), three cases,
    every layer."""
    from psm_amd import UNetSurrogate
    monkeypatch.setenv("PSM_UNET_PAIR_MIN", "1")
    for lv in (0, 1):
        assert (ny >> lv) % 30 != 0 and (nx >> lv) % 14 != 0
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=17)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=70 + k, cx=0.9, cy=0.85).astype(np.float32) for k in range(3)])   # obstacle at the corner
    with UNetSurrogate(W, ny, nx, max_cases=3, precision="bf16", keep_activations=True) as net:
        out = net.forward(grids)
        for k in range(3):
            ref, acts = uo.unet_forward(grids[k], W, return_all=True, precision="bf16")
            for i in range(len(specs) - 1):
                a = net.activation(i, 3)[k]
                err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
                assert err <= 1e-2, (specs[i].name, err)
                # the partial tiles themselves: last rows / columns of the layer
                edge = np.concatenate([(a - acts[i])[-6:].ravel(), (a - acts[i])[:, -6:].ravel()])
                scale = np.abs(acts[i]).max()
                assert np.abs(edge).max() <= 4e-2 * max(scale, 1e-6), (specs[i].name, np.abs(edge).max(), scale)
            assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,widths,c_out", [(4, (16, 32), 1), (3, (16, 32, 64), 2), (4, (16, 32, 48), 1)])
def test_gpu_unet_bf16_pairs_other_shapes(c_in, widths, c_out, monkeypatch):
    """Fused level pairs on other networks: the 4-channel stem (pressureSM_Poisson's image), two head outputs, a 48-channel
    level under the 32-channel pair (its upsample source then has 48 channels: not a multiple of 32, so that pair falls back
    to two launches while the others stay fused)."""
    from psm_amd import UNetSurrogate
    monkeypatch.setenv("PSM_UNET_PAIR_MIN", "1")
    specs = uo.unet_specs(c_in, widths, c_out)
    W = uo.he_weights(specs, seed=50 + c_in)
    m = 1 << (len(widths) - 1)
    ny, nx = 18 * m, 34 * m
    g = np.random.default_rng(c_in).standard_normal((2, ny, nx, c_in)).astype(np.float32)
    with UNetSurrogate(W, ny, nx, c_in=c_in, c_out=c_out, widths=widths, max_cases=2, precision="bf16", keep_activations=True) as net:
        out = net.forward(g)
        for k in range(2):
            ref, acts = uo.unet_forward(g[k], W, widths, return_all=True, precision="bf16")
            for i in range(len(specs) - 1):
                a = net.activation(i, 2)[k]
                err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
                assert err <= 1e-2, (specs[i].name, err)
            assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,widths,ny,nx,n", [(3, (16, 32, 64, 128, 256), 64, 96, 2), (3, (16, 32, 96, 160), 40, 72, 3), (4, (16, 32, 64, 96), 48, 80, 1)])
def test_gpu_unet_bf16_in_workgroup_k_split(c_in, widths, ny, nx, n, monkeypatch):
    """The in-workgroup K split of the generic conv kernel (psm_conv3x3_kernel<..., KW = 2>: eight waves, the two halves of a workgroup's
    channel chunks on waves 0-3 / 4-7, accumulators summed through LDS), FORCED on every eligible layer of two or more chunks
    (PSM_UNET_KW=-2; the rule takes four or more chunks and launches of at most one workgroup per CU): even chunk counts, odd ones
    (96- and 160-channel levels: 3 and 5 chunks, upsample ++ skip sources of 7 / 8 chunks -- the second half then makes up a barrier),
    max-pool / same / upsample sources, tiles that straddle the image edge; layer by layer against the bf16 oracle, and the same
    field as the unsplit kernels up to bf16 rounding flips."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs(c_in, widths, 1)
    W = uo.he_weights(specs, seed=70 + len(widths))
    g = np.random.default_rng(5).standard_normal((n, ny, nx, c_in)).astype(np.float32)
    monkeypatch.setenv("PSM_UNET_KW", "-2")
    monkeypatch.setenv("PSM_UNET_FILL", "1")          # small grids: keep the 8-row tiles and no split over workgroups (the planner would spread these layers thin)
    with UNetSurrogate(W, ny, nx, c_in=c_in, c_out=1, widths=widths, max_cases=n, precision="bf16", keep_activations=True) as net:
        roles = [net.plan_info(i)[3] for i in range(len(specs))]
        assert sum(1 for r in roles if r & 8) >= 4, roles                      # the split really ran on several layers
        out = net.forward(g)
        for k in range(n):
            ref, acts = uo.unet_forward(g[k], W, widths, return_all=True, precision="bf16")
            for i in range(len(specs) - 1):
                a = net.activation(i, n)[k]
                err = np.linalg.norm(a - acts[i]) / max(np.linalg.norm(acts[i]), 1e-12)
                assert err <= 1e-2, (specs[i].name, roles[i], err)
            # (white-noise images: more activations sit near a bf16 rounding boundary than on the smooth channel grids -- the unsplit
            # kernels are 1.0-1.1e-2 from the oracle on the 96 / 160-channel network too; tests/measure/soak_unet.py uses the same 2e-2)
            assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 2e-2
    monkeypatch.setenv("PSM_UNET_KW", "0")
    with UNetSurrogate(W, ny, nx, c_in=c_in, c_out=1, widths=widths, max_cases=n, precision="bf16") as net:
        assert not any(net.plan_info(i)[3] & 8 for i in range(len(specs)))
        off = net.forward(g)
    assert np.linalg.norm(out - off) / np.linalg.norm(off) <= 1e-2           # measured: 1e-9 ... 6e-3 (float32 summation order -> bf16 rounding flips)


@pytest.mark.gpu
def test_gpu_unet_bf16_k_split_rule_at_eight_cases():
    """The planner's rule on the bench shape (8 cases of 256 x 256, bf16): the long layers whose launch has at most one workgroup per
    CU take the in-workgroup split -- enc4b (eight chunks, as 8-row x 16-channel tiles) and dec3a (twelve chunks) among them --,
    the 64 x 64 level (512 workgroups of four waves already) and the fused pairs do not; result against the bf16 oracle."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=13)
    g = np.stack([synthetic.channel_grid(256, 256, seed=30 + k).astype(np.float32) for k in range(8)])
    with UNetSurrogate(W, 256, 256, max_cases=8, precision="bf16") as net:
        info = [net.plan_info(i) for i in range(len(specs))]
        names = [s_.name for s_ in specs]
        split = {names[i] for i in range(len(specs)) if info[i][3] & 8}
        assert {"enc4b", "dec3a"} <= split, (split, info)
        assert not ({"enc2b", "dec2a", "dec2b", "enc0a", "enc0b", "dec0a", "dec0b"} & split), split
        assert info[names.index("enc4b")][:2] == [8, 1], info[names.index("enc4b")]
        out = net.forward(g)
    for k in (0, 7):
        ref = uo.unet_forward(g[k], W, precision="bf16")
        assert np.linalg.norm(out[k] - ref) / np.linalg.norm(ref) <= 1e-2


def test_bf16_rounding_helper():
    x = np.array([1.0, 1.00390625, 1.0078125, -3.1415927, 0.0], np.float32)       # 1 + 2^-8 ties to even -> 1.0
    np.testing.assert_array_equal(uo.bf16_round(x), np.array([1.0, 1.0, 1.0078125, -3.140625, 0.0], np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("c_in,widths,c_out", [(1, (16, 32), 1), (5, (16, 32, 48), 3), (9, (16, 32), 2), (8, (32, 64), 1), (16, (16, 16), 16)])
def test_gpu_unet_channel_count_corner_cases(c_in, widths, c_out):
    """Stem variants (K = 9*c_in flattened for c_in <= 7, scalar-load staging for other counts that are not multiples
    of 4, plain chunks otherwise), fused and separate 1x1 head (first width 16 vs 32), several head outputs."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs(c_in, widths, c_out)
    W = uo.he_weights(specs, seed=100 + c_in)
    m = 1 << (len(widths) - 1)
    ny, nx = 12 * m, 20 * m
    g = np.random.default_rng(c_in).standard_normal((2, ny, nx, c_in)).astype(np.float32)
    with UNetSurrogate(W, ny, nx, c_in=c_in, c_out=c_out, widths=widths, max_cases=2) as net:
        out = net.forward(g)
        for k in range(2):
            ref, acts = uo.unet_forward(g[k], W, widths, return_all=True)
            a0 = net.activation(0, 2)[k]
            assert np.abs(a0 - acts[0]).max() <= 2e-5 * np.abs(acts[0]).max()
            assert np.abs(out[k] - ref).max() <= 1e-4 * np.abs(ref).max()


def _write_keras_conv_h5(path, W):
    import h5write
    tree = {}
    for k, (w, b) in enumerate(W):
        name = "conv2d" if k == 0 else f"conv2d_{k}"
        tree[name] = {name: {"kernel:0": w, "bias:0": b}}
    # more than 8 layers: one symbol node holds 8 members, so nest the groups two levels deep
    names = sorted(tree, key=lambda s: int(s.split("_")[1]) if "_" in s else 0)
    h5write.write_h5(path, {f"part{j}": {n: tree[n] for n in names[8 * j:8 * j + 8]} for j in range((len(names) + 7) // 8)})


def test_keras_conv_reader_and_layout_inference(tmp_path):
    from psm_amd import formats
    widths = (16, 32, 48)
    W = uo.he_weights(uo.unet_specs(4, widths, 2), seed=3)
    p = str(tmp_path / "unet.h5")
    _write_keras_conv_h5(p, W)
    R = formats.read_keras_conv_weights(p)
    assert len(R) == len(W) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(R, W))
    assert formats.unet_layout_from_weights(R) == (4, widths, 2)
    with pytest.raises(ValueError):
        formats.unet_layout_from_weights(R[:-2])


@pytest.mark.gpu
def test_gpu_unet_from_keras_h5(tmp_path):
    from psm_amd import UNetSurrogate
    widths = (16, 32, 48)
    W = uo.he_weights(uo.unet_specs(4, widths, 2), seed=3)
    p = str(tmp_path / "unet.h5")
    _write_keras_conv_h5(p, W)
    g = np.random.default_rng(2).standard_normal((32, 48, 4)).astype(np.float32)
    with UNetSurrogate.from_keras_h5(p, 32, 48) as net:
        out = net.forward(g)[0]
    ref = uo.unet_forward(g, W, widths)
    assert np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,ny,nx,n", [("bf16", 512, 512, 1), ("f32", 256, 256, 1), ("bf16", 96, 160, 2)])
def test_gpu_unet_autotuned_plan_keeps_the_result(precision, ny, nx, n):
    """psm_unet_autotune re-plans the split-K depth layer by layer from measurements: whatever it chooses, the field is the
    un-tuned plan's up to float32 summation order (bf16: rounding flips) and matches the oracle."""
    from psm_amd import UNetSurrogate
    W = uo.he_weights(uo.unet_specs(), seed=11)
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=90 + k, noise=0.05).astype(np.float32) for k in range(n)])
    with UNetSurrogate(W, ny, nx, max_cases=n, precision=precision) as net:
        plain = net.forward(grids)
        before = [int(net.lib.psm_unet_ksplit(net.h, i)) for i in range(len(net.shapes))]
    with UNetSurrogate(W, ny, nx, max_cases=n, precision=precision, autotune=True) as net:
        tuned = net.forward(grids)
        t = net.autotuned
    assert t["us_after"] <= t["us_before"] * 1.01 and len(t["ksplit"]) == 19
    # splits are halved where that pays; a measured tile choice re-runs the split rule for the tile that was chosen
    assert all(1 <= a <= 8 for a in t["ksplit"]) and len(before) == 19
    tol = 1e-2 if precision == "bf16" else 1e-5
    assert np.linalg.norm(tuned - plain) / np.linalg.norm(plain) <= tol
    ref = uo.unet_forward(grids[0], W, precision=precision)
    assert np.linalg.norm(tuned[0] - ref) / np.linalg.norm(ref) <= (1e-2 if precision == "bf16" else 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_gpu_unet_autotuned_plan_can_be_replayed_bit_identically(precision):
    """The autotuner's choices are run-dependent (a 1 % rule on measured medians); psm_unet_get_choices / set_choices let a
    second handle replay them without measuring: same plan, bit-identical field.  Bad choices are rejected and leave the
    plan as it was."""
    from psm_amd import UNetSurrogate
    W = uo.he_weights(uo.unet_specs(), seed=17)
    ny, nx = 256, 256
    grid = synthetic.channel_grid(ny, nx, seed=5, noise=0.05).astype(np.float32)[None]
    with UNetSurrogate(W, ny, nx, precision=precision, autotune=True) as net:
        tuned = net.forward(grid)
        choices = net.get_choices()
        plan = [net.plan_info(i) for i in range(len(net.shapes))]
    assert len(choices) == 19 and all(len(c) == 4 and 1 <= c[0] <= 8 for c in choices)
    with UNetSurrogate(W, ny, nx, precision=precision, choices=choices) as net:
        assert [net.plan_info(i) for i in range(len(net.shapes))] == plan and net.get_choices() == choices
        replay = net.forward(grid)
        with pytest.raises(Exception):
            net.set_choices([[9, -1, -1, -1]] * 19)
        with pytest.raises(Exception):
            net.set_choices(choices[:-1])
        assert [net.plan_info(i) for i in range(len(net.shapes))] == plan
        assert np.array_equal(net.forward(grid), replay)
    assert np.array_equal(replay, tuned)


@pytest.mark.gpu
def test_gpu_unet_x6_layers_keep_float32_accuracy(monkeypatch):
    """float32 mode: layers with at least 64 input channels and an 8-row tile run their contractions on the bf16 matrix pipe
    with operands split exactly into three bf16 planes (six MFMA terms per product).  Same oracle, same float32 tolerances,
    layer by layer; PSM_UNET_X6=0 plans no such layer and gives the same field up to float32 summation order."""
    from psm_amd import UNetSurrogate
    specs = uo.unet_specs()
    W = uo.he_weights(specs, seed=13)
    ny, nx, n = 256, 256, 4                     # enough workgroups for 8-row tiles on the 64^2 level and the wide decoder layers
    grids = np.stack([synthetic.channel_grid(ny, nx, seed=40 + k, noise=0.05).astype(np.float32) for k in range(n)])
    with UNetSurrogate(W, ny, nx, max_cases=n) as net:
        roles = [net.plan_info(i)[3] for i in range(len(specs))]
        x6 = [i for i, r in enumerate(roles) if r & 4]
        assert len(x6) >= 4 and all(specs[i].c_in >= 64 and net.plan_info(i)[0] == 8 for i in x6), roles
        out = net.forward(grids)
        for k in (0, n - 1):
            ref, acts = uo.unet_forward(grids[k], W, return_all=True)
            for i in range(len(specs) - 1):
                a = net.activation(i, n)[k]
                assert np.abs(a - acts[i]).max() <= 2e-5 * max(np.abs(acts[i]).max(), 1e-6), (specs[i].name, i in x6)
            assert np.abs(out[k] - ref).max() <= 1e-4 * np.abs(ref).max()
    monkeypatch.setenv("PSM_UNET_X6", "0")
    with UNetSurrogate(W, ny, nx, max_cases=n) as net:
        assert not any(net.plan_info(i)[3] & 4 for i in range(len(specs)))
        plain = net.forward(grids)
    assert np.linalg.norm(out - plain) / np.linalg.norm(plain) <= 1e-5
