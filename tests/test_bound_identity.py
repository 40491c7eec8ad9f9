"""The algebra behind psm_bind_geometry, in NumPy on the oracle's own quantities (no GPU): every masked strip sum of the
decoded blocks -- the only way the decoded values enter the block-offset chain (SM_call.py:233-316, Eval_dual_Dense_
onlycil.py:300-340, python_module.py:391-445) -- is linear in the network output, with coefficients that depend on the
flow-cell masks and the model only:

    sum_{cells of the strip that are flow cells} decoded[b]  =  scale * (res_inv[b] . G + M)
    G[k] = sum mask * comp_out[k][cell],   M = sum mask * mean_out[cell]

and, through the linear head layer (res_inv = (act @ W4 + b4) * sa + sb), in the last hidden activation:

    ... = scale * (act[b] . g2 + c2),   g2 = W4 @ (sa * G),   c2 = (b4 * sa + sb) . G + M."""
import numpy as np

from oracle import psm_oracle as orc
from psm_amd import synthetic
from test_oracle_golden import oracle_model


def hidden_activation(model, x_in):
    h = np.asarray(x_in, np.float32)
    for W, b in model.weights[:-1]:
        h = np.maximum(h @ W + b, np.float32(0))
    return h


def test_strip_sums_are_linear_in_the_network_output_and_in_the_last_hidden_activation():
    for variant in ("deltas", "gradp"):
        model = synthetic.make_model(variant, p_in=12, p_out=10)
        om = oracle_model(model)
        grid = synthetic.channel_grid(256, 256, seed=5).astype(np.float64)
        sol = orc.solve_grid(grid, om)
        lay = orc.block_layout(variant, 256, 256, model.S, om.overlap())
        xb = orc.extract_blocks(grid, lay, model.c_in)
        S, C, ov = model.S, model.c_out, lay.ov
        res_inv = om.scaler.inv(sol.res.astype(np.float64))                     # [B, P_o]
        act = hidden_activation(model, sol.x_input).astype(np.float64)          # [B, 512]
        W4, b4 = (np.asarray(a, np.float64) for a in model.weights[-1])
        sa = np.broadcast_to(np.asarray(model.out_b if model.scaler_kind != "max_abs" else model.out_a, np.float64), (model.p_out,))
        sb = np.broadcast_to(np.asarray(model.out_a if model.scaler_kind != "max_abs" else 0.0, np.float64), (model.p_out,))
        np.testing.assert_allclose((act @ W4 + b4) * sa + sb, res_inv, rtol=0, atol=2e-5 * np.abs(res_inv).max())
        comp = np.asarray(model.comp_out, np.float64).reshape(model.p_out, S, S, C)
        mean = np.asarray(model.mean_out, np.float64).reshape(S, S, C)
        rng = np.random.default_rng(0)
        for trial in range(12):
            b = int(rng.integers(1, lay.B))
            data_blk = b if trial % 2 == 0 else b - 1        # the previous block's strip under THIS block's mask (SMD:235)
            f = int(rng.integers(0, C))
            rect = [(slice(0, S), slice(S - ov, S)), (slice(0, ov), slice(0, S)), (slice(S - ov, S), slice(0, S)),
                    (slice(0, S), slice(0, ov))][trial % 4]
            mask = (xb[b][..., model.sdf_ch] != 0)[rect]
            direct = (sol.block_pred[data_blk][..., f][rect] * mask).sum()
            G = (comp[:, :, :, f][(slice(None),) + rect] * mask).sum(axis=(1, 2))
            M = (mean[..., f][rect] * mask).sum()
            via_res = model.out_scale * (res_inv[data_blk] @ G + M)
            g2, c2 = W4 @ (sa * G), (b4 * sa + sb) @ G + M
            via_act = model.out_scale * (act[data_blk] @ g2 + c2)
            scale = max(abs(direct), 1.0)
            assert abs(via_res - direct) <= 1e-9 * scale
            assert abs(via_act - direct) <= 2e-5 * scale * max(1.0, mask.sum() ** 0.5)
