"""bench.py keeps the driver's contract: one JSON line with the agreed keys (run as a subprocess, short)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}
ROOF = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_bench_line():
    d = run_bench()
    assert KEYS <= set(d) and ROOF <= set(d["roofline"])
    assert d["metric"].startswith("pressure-solves/sec") and d["unit"] == "solves/s" and d["n_gpus"] == 1
    assert d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 1000 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["geometry"].startswith("bound once")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert d["l2_vs_oracle"] < 1e-5


@pytest.mark.gpu
def test_general_path_and_conv_bench_lines():
    d = run_bench("--no-bind", "--no-cpu-baseline")
    assert d["config"]["geometry"].startswith("general path") and "cpu_baseline" not in d
    u = run_bench("--workload", "unet", "--no-cpu-baseline")
    assert KEYS <= set(u) and u["roofline"]["bound"] == "mfma" and u["roofline"]["unit"] == "TFLOP/s" and u["value"] > 100
