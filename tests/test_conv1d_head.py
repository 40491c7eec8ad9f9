"""The reference's conv1D_PCA network (NNs.py:75-124; architecture 'conv1D' of utils.define_model_arch, utils.py:452-454) as
a loadable model: Conv1D stack over the scaled PCA coefficients, Flatten, Dense head.

Pinned by tests/golden/deltas_conv1d_256x256.npz: the network there was BUILT BY THE REFERENCE'S OWN FUNCTION (layer
list, filter counts, kernel size, padding, activations, Flatten, head -- tests/golden/make_golden.py executes
`define_model_arch('conv1D')` and `conv1D_PCA(...)` with NumPy layer stand-ins) and run inside the reference's timeStep
statements.  Tolerances as for the Dense stacks: network output rel-L2 <= 1e-5, fields <= 2e-4 * max|field|."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate, _lib, formats, synthetic
from test_gpu_parity import check_against_oracle, oracle_model, rel_l2


def test_golden_network_is_the_reference_architecture():
    grid, model = cases.build("deltas_conv1d_256x256")
    assert [K.shape for K, _ in model.conv1d] == [(3, 1, 128), (3, 128, 64), (3, 64, 32), (3, 32, 16), (3, 16, 32), (3, 32, 64), (3, 64, 128)]
    assert [W.shape for W, _ in model.weights] == [(32 * 128, 32)]
    assert synthetic.CONV1D_WIDTHS == [K.shape[2] for K, _ in model.conv1d]


def test_oracle_conv1d_is_tensorflow_same_padding():
    """Against torch's conv1d (cross-correlation, explicit padding): odd and even kernel sizes -- TensorFlow's 'same' puts
    (k - 1) // 2 zeros in front and the rest behind."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for k in (1, 2, 3, 4, 5):
        x = rng.standard_normal((5, 37)).astype(np.float32)
        convs = [((rng.standard_normal((k, 1, 6)) * 0.5).astype(np.float32), rng.standard_normal(6).astype(np.float32)),
                 ((rng.standard_normal((k, 6, 3)) * 0.5).astype(np.float32), rng.standard_normal(3).astype(np.float32))]
        got = orc.conv1d_forward(x, convs)
        h = torch.from_numpy(x)[:, None, :]
        for K, b in convs:
            front = (k - 1) // 2
            h = F.relu(F.conv1d(F.pad(h, (front, k - 1 - front)), torch.from_numpy(K).permute(2, 1, 0).contiguous(), torch.from_numpy(b)))
        want = h.permute(0, 2, 1).reshape(5, -1).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_keras_reader_finds_the_conv1d_head(tmp_path):
    import h5write
    convs, dense = synthetic.he_conv1d_head(24, [8, 4, 8], 16, seed=5)
    tree = {}
    for k, (K, b) in enumerate(convs):
        n = "conv1d" if k == 0 else f"conv1d_{k}"
        tree[n] = {n: {"kernel:0": K, "bias:0": b}}
    tree["dense"] = {"dense": {"kernel:0": dense[0][0], "bias:0": dense[0][1]}}
    p = str(tmp_path / "conv1d.h5")
    h5write.write_h5(p, {"model_weights": tree})
    c2, d2 = formats.read_keras_conv1d_head(p)
    assert len(c2) == 3 and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(c2, convs))
    assert np.array_equal(d2[0][0], dense[0][0])
    p2 = str(tmp_path / "dense.h5")
    h5write.write_keras_dense(p2, synthetic.he_dense_stack(8, [16], 4, 1))
    c3, d3 = formats.read_keras_conv1d_head(p2)
    assert c3 == [] and len(d3) == 2


@pytest.mark.gpu
def test_gpu_conv1d_golden():
    name = "deltas_conv1d_256x256"
    grid, model = cases.build(name)
    gold = cases.load_golden(name)
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, 256, 256) as sur:
        fields = sur.solve(g32, out_scale=[model.out_scale])[0]
        sol = orc.solve_grid(g32.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, g32, model, sol)
        assert not sur.bind_geometry(g32)                       # no hidden Dense layer: the general path stays
        again = sur.solve(g32, out_scale=[model.out_scale])[0]
    np.testing.assert_array_equal(again, fields)
    ref = gold["fields"]
    assert np.abs(fields - ref).max() <= 2e-4 * np.abs(ref).max() and rel_l2(fields, ref) <= 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("p_in,filters,k,n_cases", [(45, [16, 8], 3, 1), (33, [5, 7, 3], 2, 2), (128, [128, 64, 32, 16, 32, 64, 128], 3, 3), (20, [4], 5, 1)])
def test_gpu_conv1d_shapes(p_in, filters, k, n_cases):
    """Coefficient counts that are not multiples of 4, odd filter counts, even and long kernels, the reference's full-size
    architecture on 128 coefficients (Flatten 16384 -> Dense), case batches; and a hidden Dense layer behind the Flatten."""
    model = synthetic.make_model("deltas", p_in=p_in, p_out=24, seed_pca=500 + p_in)
    model.conv1d, model.weights = synthetic.he_conv1d_head(p_in, filters, 24, seed=p_in, kernel_size=k)
    if p_in == 45:                                              # Flatten -> Dense(64, relu) -> Dense: binds like a Dense stack
        W0 = model.weights[0][0]
        rng = np.random.default_rng(1)
        model.weights = [((rng.standard_normal((W0.shape[0], 64)) / np.sqrt(W0.shape[0])).astype(np.float32), np.zeros(64, np.float32)),
                         ((rng.standard_normal((64, 24)) / 8).astype(np.float32), np.zeros(24, np.float32))]
    grids = synthetic.random_obstacle_cases(n_cases, 256, 256, seed=p_in).astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=n_cases) as sur:
        fields = sur.solve(grids)
        for c in range(n_cases):
            sol = orc.solve_grid(grids[c].astype(np.float64), oracle_model(model))
            check_against_oracle(sur, grids[c], model, sol, n_cases=n_cases, case=c)
            assert np.abs(fields[c] - sol.fields).max() <= 1e-4 * np.abs(sol.fields).max()
        if p_in == 45:
            assert sur.bind_geometry(grids)
            b = sur.solve(grids)
            assert np.abs(b - fields).max() <= 2e-5 * np.abs(fields).max()


@pytest.mark.gpu
def test_gpu_conv1d_errors():
    model = synthetic.make_model("deltas", p_in=16, p_out=8)
    model.conv1d, model.weights = synthetic.he_conv1d_head(16, [4, 8], 8, seed=2)
    with pytest.raises(_lib.PsmError):
        GridSurrogate(model, 256, 256, precision="bf16")       # float32 only
    bad = synthetic.make_model("deltas", p_in=16, p_out=8)
    bad.conv1d, bad.weights = synthetic.he_conv1d_head(16, [4, 8], 8, seed=2)
    bad.conv1d[1] = (bad.conv1d[1][0][:, :3, :].copy(), bad.conv1d[1][1])      # does not chain
    with pytest.raises(_lib.PsmError):
        GridSurrogate(bad, 256, 256)
