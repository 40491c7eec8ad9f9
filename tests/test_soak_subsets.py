"""Small fixed-seed runs of the randomised soaks under tests/measure/ (the long runs are recorded under profiles/): each script
draws configurations, compares the HIP path with the CPU oracle (or with the synchronous entry) and stops at the first
difference.  Run in-process (one GPU process per test session)."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,args", [
    ("soak.py", ["24", "12"]),                 # surrogate path: general / bound / batched / next step against the oracle
    ("soak_next_rows.py", ["25", "5"]),        # integrate_gradp, gaussian_filter, Poisson features
    ("soak_ring.py", ["8", "3"]),              # asynchronous host-buffer entries against the synchronous one
])
def test_soak_subset(script, args, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script] + args)
    runpy.run_path(os.path.join(ROOT, "tests", "measure", script), run_name="__main__")
    assert "SOAK OK" in capsys.readouterr().out
