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


def test_gpus_n_starts_n_ranks_without_a_launcher():
    """`bench.py --gpus 2` with no torchrun environment spawns its two ranks itself (gloo, no GPU work: --dry-run)
    and reports the world size the process group saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PSM_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_reported"] == 2 and d["dry_run"] is True and d["steps"] == 5


def test_a_dying_rank_ends_the_job_within_seconds(tmp_path):
    """Three self-launched gloo ranks, rank 2 exits before the rendezvous: the launcher must not wait for the other two
    (which sit in the rendezvous) -- it stops them, returns non-zero quickly and keeps every rank's stderr."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSM_BENCH_BACKEND="gloo", PSM_BENCH_FAIL_RANK="2", PSM_BENCH_LOGDIR=str(tmp_path))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    took = time.time() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-1000:])
    assert took < 30.0, took                      # dominated by three interpreter + torch start-ups, not by a rendezvous timeout
    assert "rank 2 exited with 3" in out.stderr and "PSM_BENCH_FAIL_RANK" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    for r in range(3):
        assert (tmp_path / f"bench_rank{r}.err").exists()
    assert "PSM_BENCH_FAIL_RANK" in (tmp_path / "bench_rank2.err").read_text()


def test_launcher_deadline(tmp_path):
    """PSM_BENCH_TIMEOUT bounds the whole job: ranks that never finish are stopped and the launcher returns 124."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSM_BENCH_BACKEND="gloo", PSM_BENCH_LOGDIR=str(tmp_path), PSM_BENCH_TIMEOUT="0.5")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 124 and "PSM_BENCH_TIMEOUT" in out.stderr


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PSM_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "world size" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_default_bench_line():
    d = run_bench()
    assert KEYS <= set(d) and ROOF <= set(d["roofline"])
    assert d["metric"].startswith("pressure-solves/sec") and d["unit"] == "solves/s" and d["n_gpus"] == 1
    assert d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 1000 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["geometry"].startswith("bound once")
    assert "device-resident" in d["config"]["value_is"] and d["config"]["degenerate"].startswith("build-defined skip")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the roofline kernel is the one with the largest measured time of the instrumented pass
    top = max(r["kernels"], key=lambda k: k["avg_us"] * k["launches_per_solve"])
    assert r["kernel"] == top["name"] and len(r["kernels"]) >= 6 and all(k["launches_per_solve"] == 1 for k in r["kernels"])
    assert sum("#layer" in k["name"] for k in r["kernels"]) == 3          # dense 2, dense 3, head (layer 1 is fused with the slab reduction)
    # the H2D / D2H-inclusive solve, and BASELINE configs[3]
    e = d["end_to_end"]
    assert d["value_end_to_end"] > 1000 and e["matches_device_resident_result"] is True
    assert {"sync_pageable", "ring_pageable_depth4", "ring_registered_depth4", "ring_zero_copy_depth4"} <= set(e["solves_per_s_per_rank"])
    cbat = d["case_batch"]
    assert "configs[3]" in cbat["workload"] and cbat["cases_per_step_per_gpu"] == 8 and cbat["value"] > 1000
    assert cbat["gathered_shape"] == [8, 256, 256, 1] and cbat["total_cases"] == 8 and cbat["guard_trips"] == 0
    assert cbat["roofline"]["bound"] in ("mfma", "hbm") and cbat["roofline"]["frac"] > 0 and cbat["l2_vs_oracle"] < 1e-5
    assert d["value_device_resident"] == d["value"] and len(d["devices"]) == 1 and d["config"]["guard_trips"] == 0
    assert e["hw_queues"]["hip_initialised_before_it_was_set"] is False
    legs = d["legs"]
    assert set(legs) == {"config2", "config4", "unet", "unet8", "unet8_bf16", "unet512_bf16"}
    for name, leg in legs.items():
        assert leg["ms_per_step"] > 0 and leg["value"] > 100 and leg["l2_vs_oracle"] < (2e-2 if leg["dtype"] == "bf16" else 1e-5), name
        assert ROOF <= set(leg["roofline"]) and 0 < leg["roofline"]["frac"] < 1, name
    assert legs["unet512_bf16"]["grid"] == [512, 512] and legs["unet512_bf16"]["cpu_baseline"]["value"] > 0
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert d["l2_vs_oracle"] < 1e-5


@pytest.mark.gpu
def test_general_path_and_conv_bench_lines():
    d = run_bench("--no-bind", "--no-cpu-baseline")
    assert d["config"]["geometry"].startswith("general path") and "cpu_baseline" not in d
    u = run_bench("--workload", "unet", "--no-cpu-baseline")
    assert KEYS <= set(u) and u["roofline"]["bound"] in ("mfma", "hbm") and u["roofline"]["whole_pass"]["achieved_TFLOPs"] > 1 and u["value"] > 100
