"""The round-6 fast paths have a switch each; the forms they replace stay in the library -- for fields of
4 GiB or more (branch-predicated paste stores) and for models whose Dense chain cannot be packed -- and must give the same fields.
The switches are read once per process, so each form runs in a process of its own; the arithmetic is identical, the fields must be too."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import psm_amd
from psm_amd import synthetic
out = {}
for variant, n in (("deltas", 20), ("gradp", 1), ("deltas", 1)):
    model = synthetic.make_model(variant, p_in=64, p_out=64, seed_pca=5, seed_w=6)
    grids = synthetic.random_obstacle_cases(n, 256, 256, seed=11).astype(np.float32)
    with psm_amd.GridSurrogate(model, 256, 256, max_cases=n) as sur:
        assert sur.bind_geometry(grids, n_cases=n)
        out[f"{variant}_{n}"] = sur.solve(grids)
np.savez(sys.argv[1], **out)
'''


@pytest.mark.gpu
def test_replaced_forms_give_the_same_fields(tmp_path):
    def run(tag, **env):
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return np.load(path)
    base = run("default")
    assert all(np.isfinite(base[k]).all() for k in base.files)
    branchy = run("branches", PSM_PASTE_BUFFER_STORES="0")          # exec-mask branch + 64-bit address per pasted value
    rows = run("rows", PSM_DENSE_PACKED="0")                        # hidden activations of the 20-case batch (180 block rows) as rows
    for k in base.files:
        np.testing.assert_array_equal(branchy[k], base[k], err_msg=k)
        np.testing.assert_array_equal(rows[k], base[k], err_msg=k)
