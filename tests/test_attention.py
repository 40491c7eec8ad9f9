"""The reference's densePCA_attention network (NNs.py:40-72; architecture 'MLP_attention' of utils.define_model_arch,
utils.py:455-457) as a loadable model: Dense -> MultiHeadAttention over a sequence of length 1 -> LayerNormalization ->
(Dense + residual -> LayerNormalization) x (n_layers - 1) -> Dense head.

Pinned by tests/golden/deltas_attention_256x256.npz: the network there was BUILT BY THE REFERENCE'S OWN FUNCTION (which
layers exist, their order, what the attention is called with, where the residuals are -- tests/golden/make_golden.py executes
`define_model_arch('MLP_attention')` and `densePCA_attention(...)` with NumPy stand-ins for the Keras layers) and run inside
the reference's timeStep statements.  Tolerances as for the Dense stacks: network output rel-L2 <= 1e-5, fields <= 2e-4 * max."""
import numpy as np
import pytest

import cases
from oracle import psm_oracle as orc
from psm_amd import GridSurrogate, _lib, synthetic
from test_gpu_parity import check_against_oracle, oracle_model, rel_l2

NAME = "deltas_attention_256x256"


def test_golden_network_is_the_reference_architecture():
    grid, model = cases.build(NAME)
    assert [W.shape for W, _ in model.weights] == [(32, 512), (512, 512), (512, 512), (512, 32)]
    a = model.attention
    assert a["Wv"].shape == (512, 8, 64) and a["Wo"].shape == (8, 64, 512) and len(a["ln"]) == 3 and a["eps"] == 1e-3
    assert synthetic.ARCHS["MLP_attention"] == [512] * 3


def test_oracle_attention_is_a_real_multi_head_attention():
    """The oracle's MultiHeadAttention restatement against torch's (packed in-projection, per-head scaled dot product, softmax
    over the keys, output projection) on sequences LONGER than one -- so that what collapses at length 1 is the real layer."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    d, H, kd, B, T, S = 48, 4, 12, 3, 5, 7                      # torch needs d == H * kd
    att = {k: (rng.standard_normal(sh) * 0.3).astype(np.float32) for k, sh in
           dict(Wq=(d, H, kd), bq=(H, kd), Wk=(d, H, kd), bk=(H, kd), Wv=(d, H, kd), bv=(H, kd), Wo=(H, kd, d), bo=(d,)).items()}
    q_in, kv_in = rng.standard_normal((B, T, d)).astype(np.float32), rng.standard_normal((B, S, d)).astype(np.float32)
    got = orc.multi_head_attention(q_in, kv_in, att)
    t = torch.from_numpy
    in_w = torch.cat([t(att["W" + k].reshape(d, H * kd).T.copy()) for k in "qkv"])
    in_b = torch.cat([t(att["b" + k].reshape(-1)) for k in "qkv"])
    ref, _ = F.multi_head_attention_forward(t(q_in).transpose(0, 1), t(kv_in).transpose(0, 1), t(kv_in).transpose(0, 1), d, H, in_w, in_b,
                                            None, None, False, 0.0, t(att["Wo"].reshape(H * kd, d).T.copy()), t(att["bo"]), need_weights=False)
    ref = ref.transpose(0, 1).numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def test_oracle_layer_normalization_is_torch_layer_norm():
    import torch
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((7, 100)) * 3 + 1.5).astype(np.float32)
    g, b = rng.standard_normal(100).astype(np.float32), rng.standard_normal(100).astype(np.float32)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (100,), torch.from_numpy(g), torch.from_numpy(b), eps=1e-3).numpy()
    assert np.abs(orc.layer_normalization(x, g, b, 1e-3) - ref).max() <= 1e-5


def test_length_one_attention_is_affine_and_ignores_query_and_key():
    """What psm_set_attention relies on: over the reference's sequence of length 1 (NNs.py:54) the softmax is exactly 1, so the
    block equals (x . Wv + bv) . Wo + bo whatever the query / key projections are."""
    _, model = cases.build(NAME)
    a = dict(model.attention)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((9, 512)).astype(np.float32)
    out = orc.multi_head_attention(x[:, None], x[:, None], a)[:, 0]
    b = dict(a, Wq=a["Wq"] * 37.0, bq=a["bq"] - 5.0, Wk=-a["Wk"], bk=a["bk"] * 0.0)
    np.testing.assert_array_equal(orc.multi_head_attention(x[:, None], x[:, None], b)[:, 0], out)
    HV = 8 * 64
    folded = (x.astype(np.float64) @ a["Wv"].reshape(512, HV) + a["bv"].reshape(-1)) @ a["Wo"].reshape(HV, 512) + a["bo"]
    assert np.abs(out - folded).max() <= 2e-6 * np.abs(folded).max()


def test_keras_reader_finds_the_attention_block(tmp_path):
    """A Keras-layout HDF5 file of densePCA_attention (Dense layers, one MultiHeadAttention with its four EinsumDense
    sub-layers, LayerNormalization layers in creation order) -> the model dict the loader hands to the library; a plain Dense
    file has no attention part."""
    import h5write
    from psm_amd import formats
    dense = synthetic.he_dense_stack(12, [32, 32, 32], 6, seed=3)
    att = synthetic.he_attention_block([32, 32, 32], seed=4, n_heads=2, key_dim=8)
    tree = {}
    for k, (W, b) in enumerate(dense):
        n = "dense" if k == 0 else f"dense_{k}"
        tree[n] = {n: {"kernel:0": W, "bias:0": b}}
    m = "multi_head_attention"
    tree[m] = {m: {"query": {"kernel:0": att["Wq"], "bias:0": att["bq"]}, "key": {"kernel:0": att["Wk"], "bias:0": att["bk"]},
                   "value": {"kernel:0": att["Wv"], "bias:0": att["bv"]},
                   "attention_output": {"kernel:0": att["Wo"], "bias:0": att["bo"]}}}
    for k, (g, b) in reversed(list(enumerate(att["ln"]))):         # file order is not creation order
        n = "layer_normalization" if k == 0 else f"layer_normalization_{k}"
        tree[n] = {n: {"gamma:0": g, "beta:0": b}}
    p = str(tmp_path / "attention.h5")
    h5write.write_h5(p, {"model_weights": tree})
    got = formats.read_keras_attention(p)
    assert got["eps"] == 1e-3 and len(got["ln"]) == 3
    for k in ("Wq", "bq", "Wk", "bk", "Wv", "bv", "Wo", "bo"):
        np.testing.assert_array_equal(got[k], att[k])
    for (g, b), (g0, b0) in zip(got["ln"], att["ln"]):
        np.testing.assert_array_equal(g, g0); np.testing.assert_array_equal(b, b0)
    d2 = formats.read_keras_dense_weights(p)
    assert [W.shape for W, _ in d2] == [W.shape for W, _ in dense]
    # the two network functions agree on what was read
    x = np.random.default_rng(0).standard_normal((5, 12)).astype(np.float32)
    np.testing.assert_array_equal(orc.mlp_attention_forward(x, d2, got), orc.mlp_attention_forward(x, dense, att))
    p2 = str(tmp_path / "dense.h5")
    h5write.write_keras_dense(p2, dense)
    assert formats.read_keras_attention(p2) is None


@pytest.mark.gpu
def test_gpu_attention_golden_general_bound_and_batch():
    grid, model = cases.build(NAME)
    gold = cases.load_golden(NAME)
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, 256, 256, max_cases=3) as sur:
        fields = sur.solve(g32, out_scale=[model.out_scale])[0]
        sol = orc.solve_grid(g32.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, g32, model, sol)
        ref = gold["fields"]
        assert np.abs(fields - ref).max() <= 2e-4 * np.abs(ref).max() and rel_l2(fields, ref) <= 5e-5
        assert sur.bind_geometry(g32)                           # the geometry-bound path (head + strip dots in one launch) is kept
        bound = sur.solve(g32, out_scale=[model.out_scale])[0]
        assert sur.guard_trips == 0 and np.abs(bound - ref).max() <= 2e-4 * np.abs(ref).max()
        sur.unbind_geometry()
        batch = np.stack([g32, g32[::-1].copy(), g32])          # case batch: rows of all cases through one LayerNormalization launch
        out = sur.solve(batch, out_scale=[model.out_scale] * 3)
        np.testing.assert_array_equal(out[0], out[2])
        assert np.abs(out[0] - ref).max() <= 2e-4 * np.abs(ref).max()
        sol1 = orc.solve_grid(batch[1].astype(np.float64), oracle_model(model))
        assert np.abs(out[1] - sol1.fields * model.out_scale / model.out_scale).max() <= 2e-4 * np.abs(sol1.fields).max()


@pytest.mark.gpu
@pytest.mark.parametrize("p,width,n_layers,variant", [(45, 96, 2, "chapter5"), (128, 512, 3, "gradp"), (24, 200, 4, "deltas")])
def test_gpu_attention_shapes(p, width, n_layers, variant):
    """Other widths (not a multiple of 32), depths and variants; head counts / dims that do not multiply to the width."""
    model = synthetic.make_model(variant, p_in=p, p_out=p, seed_pca=700 + p,
                                 weights=synthetic.he_dense_stack(p, [width] * n_layers, p, seed=p))
    model.attention = synthetic.he_attention_block([width] * n_layers, seed=p + 1, n_heads=3, key_dim=20)
    grid = synthetic.channel_grid(256, 288, seed=p).astype(np.float32)
    with GridSurrogate(model, 256, 288) as sur:
        fields = sur.solve(grid)[0]
        sol = orc.solve_grid(grid.astype(np.float64), oracle_model(model))
        check_against_oracle(sur, grid, model, sol)
        assert np.abs(fields - sol.fields).max() <= 2e-4 * np.abs(sol.fields).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n_cases", [1, 5])
def test_gpu_attention_fused_and_standalone_layernorm_agree(monkeypatch, n_cases):
    """The LayerNormalizations that feed a hidden Dense layer run inside that layer's launch (moments of its own input rows in
    the prologue, residual in the epilogue); PSM_LN_FUSE=0 launches each one on its own.  Same numbers up to float32 rounding
    of the moments; 5 cases = 45 block rows take the 16-row tiles with other row groups, 160 rows the 32-row tiles."""
    grid, model = cases.build(NAME)
    g32 = np.stack([np.roll(grid.astype(np.float32), 7 * k, axis=1) for k in range(n_cases)])
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PSM_LN_FUSE", mode)
        with GridSurrogate(model, 256, 256, max_cases=n_cases) as sur:
            outs[mode] = sur.solve(g32, out_scale=[model.out_scale] * n_cases)
            res = sur.stage("res", n_cases)
            outs[mode + "res"] = res
    assert rel_l2(outs["1res"], outs["0res"]) <= 2e-6 and np.abs(outs["1"] - outs["0"]).max() <= 2e-5 * np.abs(outs["0"]).max()
    sol = orc.solve_grid(g32[n_cases - 1].astype(np.float64), oracle_model(model))
    assert np.abs(outs["1"][n_cases - 1] - sol.fields).max() <= 2e-4 * np.abs(sol.fields).max()


@pytest.mark.gpu
def test_gpu_attention_32_row_tiles_and_wide_layers():
    """More than 128 block rows (the 32-row tile of the Dense kernel: two row sets per lane in the moment prologue) and a hidden
    width above 512 (two passes over the contraction: moments from L2 instead of the operand registers)."""
    model = synthetic.make_model("deltas", p_in=40, p_out=24, seed_pca=77, weights=synthetic.he_dense_stack(40, [640] * 3, 24, seed=5))
    model.attention = synthetic.he_attention_block([640] * 3, seed=6, n_heads=5, key_dim=16)
    grids = synthetic.random_obstacle_cases(16, 256, 256, seed=12).astype(np.float32)        # 16 cases x 9 blocks = 144 rows
    with GridSurrogate(model, 256, 256, max_cases=16) as sur:
        out = sur.solve(grids)
    for c in (0, 15):
        sol = orc.solve_grid(grids[c].astype(np.float64), oracle_model(model))
        assert np.abs(out[c] - sol.fields).max() <= 2e-4 * np.abs(sol.fields).max()


@pytest.mark.gpu
def test_gpu_attention_bf16_handle():
    """bf16 operands: the oracle rounds what enters each contraction, with the attention block as the ONE folded affine layer
    the library runs; the normalisations are float32 on both sides."""
    grid, model = cases.build(NAME)
    g32 = grid.astype(np.float32)
    with GridSurrogate(model, 256, 256, precision="bf16") as sur:
        fields = sur.solve(g32, out_scale=[model.out_scale])[0]
    sol = orc.solve_grid(g32.astype(np.float64), oracle_model(model), precision="bf16")
    assert rel_l2(fields, sol.fields) <= 2e-2


@pytest.mark.gpu
def test_gpu_attention_errors():
    _, model = cases.build(NAME)
    bad = synthetic.make_model("deltas", p_in=32, p_out=32, weights=synthetic.he_dense_stack(32, [64, 48, 64], 32, seed=1))
    bad.attention = synthetic.he_attention_block([64] * 3, seed=2)
    with pytest.raises(_lib.PsmError):                          # x + attn_output needs equal widths (48 != 64)
        GridSurrogate(bad, 256, 256)
    short = synthetic.make_model("deltas", p_in=32, p_out=32, weights=model.weights)
    short.attention = dict(model.attention, ln=model.attention["ln"][:2])
    with pytest.raises(ValueError):
        GridSurrogate(short, 256, 256)
    with GridSurrogate(model, 256, 256) as sur:
        import ctypes as C
        g = np.ones(512, np.float32)
        p = g.ctypes.data_as(C.POINTER(C.c_float))
        assert sur.lib.psm_set_layernorm(sur.h, 4, 32, p, p, 1e-3, 0) != 0          # the head takes no LayerNormalization
        assert sur.lib.psm_set_layernorm(sur.h, 1, 100, p, p, 1e-3, 0) != 0         # width mismatch
        assert sur.lib.psm_set_layernorm(sur.h, 1, 512, p, p, 0.0, 0) != 0          # epsilon
        assert sur.lib.psm_set_layernorm(sur.h, 0, 512, p, p, 1e-3, 1) != 0         # residual on a 32 -> 512 layer
        assert sur.lib.psm_set_attention(sur.h, 0, 512, 8, 64, p, p, p, p) != 0     # not behind the first Dense layer


@pytest.mark.gpu
def test_gpu_evaluator_loads_an_attention_file(tmp_path):
    """`Evaluation(..., model_path=<densePCA_attention .h5>)` -- the artefact directory of tests/cases.py with its network file
    replaced by a Keras-layout attention file: load_artifacts picks up the MultiHeadAttention / LayerNormalization groups, the
    frame solved from the files equals the oracle's network on the same grid, and call_SM_main reports the block-level error."""
    import os
    import h5write
    from psm_amd import Evaluation, call_SM_main, formats
    d = str(tmp_path)
    c = cases.build_dataset_case(d)
    pc = 24
    dense = synthetic.he_dense_stack(pc, [64, 64, 64], pc, seed=9)
    att = synthetic.he_attention_block([64, 64, 64], seed=10, n_heads=4, key_dim=8)
    tree = {}
    for k, (W, b) in enumerate(dense):
        n = "dense" if k == 0 else f"dense_{k}"
        tree[n] = {n: {"kernel:0": W, "bias:0": b}}
    m = "multi_head_attention"
    tree[m] = {m: {"query": {"kernel:0": att["Wq"], "bias:0": att["bq"]}, "key": {"kernel:0": att["Wk"], "bias:0": att["bk"]},
                   "value": {"kernel:0": att["Wv"], "bias:0": att["bv"]}, "attention_output": {"kernel:0": att["Wo"], "bias:0": att["bo"]}}}
    for k, (g, b) in enumerate(att["ln"]):
        n = "layer_normalization" if k == 0 else f"layer_normalization_{k}"
        tree[n] = {n: {"gamma:0": g, "beta:0": b}}
    path = os.path.join(d, "attention.h5")
    h5write.write_h5(path, {"model_weights": tree})
    ev = Evaluation(5e-3, 128, 32, 0.95, 0.95, c["dataset_path"], path, 128, "std", artifact_dir=d)
    assert ev.artifacts.attention is not None and len(ev.artifacts.attention["ln"]) == 3
    ev.computeOnlyOnce(0)
    res = ev.timeStep(0, 1, False, False, False, False)
    model = c["model"]
    import dataclasses
    om_model = dataclasses.replace(model, weights=dense, attention=att)
    om = oracle_model(om_model)
    om.out_scale = cases.DATASET_MAXS[3] * ev.U_max_norm ** 2
    sol = orc.solve_grid(ev.grid[..., :3], om)
    assert np.abs(res - sol.fields[..., 0]).max() <= 1e-4 * np.abs(sol.fields).max()
    rep = call_SM_main(5e-3, path, 128, 0.25, 0.95, 0.95, 128, c["dataset_path"], False, "std", False, False, False, False, 1, 2, artifact_dir=d)
    assert rep["overall"]["RSME_block"] > 0 and np.isfinite(rep["overall"]["BIAS_block"])
