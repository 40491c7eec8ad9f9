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


LINE_LIMIT = 4096       # the driver keeps ~8 KB of stdout (round 3: a 42 KB line was cut and the record had parsed = null)


def run_bench(*extra, logdir=None):
    """-> (compact line as the driver sees it, full record from bench_detail.json)."""
    logdir = logdir or os.path.join(ROOT, "gpurun_out", "bench_test")
    env = dict(os.environ, PSM_BENCH_LOGDIR=str(logdir))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout) < LINE_LIMIT and len(out.stderr) < LINE_LIMIT, (len(out.stdout), len(out.stderr))
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    return json.loads(lines[0]), json.load(open(os.path.join(logdir, "bench_detail.json")))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("psm_bench_script", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "GPU_MAX_HW_QUEUES")}
    try:
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():                   # bench.py sizes thread pools through the environment: not in this process
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return m


def test_line_built_from_a_recorded_run_fits_the_driver():
    """The full record of a real run (round 3's 42 KB line, kept as a fixture) -> the line bench.py prints now: every
    contract key, roofline and cpu_baseline present, shorter than 4096 characters, strings bounded."""
    b = _bench_module()
    detail = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r03.json")))
    assert len(json.dumps(detail)) > 30000
    line = b.compact_line(detail)
    assert len(line) < LINE_LIMIT and "\n" not in line
    d = json.loads(line)
    assert KEYS <= set(d) and ROOF <= set(d["roofline"]) and {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["value"] == detail["value"] and d["ms_per_step"] == detail["ms_per_step"] and d["steps"] == detail["steps"]
    assert "configs[1]" in d["config"]["workload"] and d["dtype"] == "f32" and d["detail"] == "bench_detail.json"
    assert all(len(v) <= 160 for v in d["config"].values() if isinstance(v, str)) and len(d["cpu_baseline"]["sample"]) <= 120
    assert set(d["roofline"]) - {"traffic_source"} == {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us"}
    assert len(d["roofline"].get("traffic_source", "")) <= 48           # round 6: where `traffic` comes from stays on the line, shortened
    assert set(d["legs"]) == set(detail["legs"])
    for leg in list(d["legs"].values()) + [d["case_batch"]]:
        assert {"value", "ms_per_step", "dtype", "l2_vs_oracle", "bound", "frac", "cpu"} <= set(leg)
    # a pathological record (hundreds of legs, kilobyte strings) still yields a line the driver can keep
    fat = dict(detail, legs={f"leg{i}": detail["legs"]["unet"] for i in range(300)})
    fat["config"] = dict(detail["config"], workload="x" * 5000, geometry="y" * 5000)
    fat["devices"] = ["0000:0d:00.0"] * 64
    line = b.compact_line(fat)
    assert len(line) < LINE_LIMIT and KEYS <= set(json.loads(line))


def test_round5_keys_reach_the_line():
    """Round-5 additions to the record -- the per-solve quantiles (BASELINE.md section 2's protocol: >= 200 solves, median and the
    10th / 90th percentile), the headline's frac_pass and the shipped_case leg (the reference's deployment shape in ITS unit, ms per
    psm_solve call, DLPoissonFoam.C:111) -- are carried by the compact line, rounded, and the line still fits the driver."""
    b = _bench_module()
    detail = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_r03.json")))
    detail.update(value_p50=29123.456789, value_p10=27001.2, value_p90=30555.9, frac_pass=0.161234567,
                  per_solve_quantiles={"solves": 200, "ms_p50": 0.0343, "ms_p10": 0.0327, "ms_p90": 0.037},
                  shipped_case={"what": "x" * 400, "ms_per_call": 0.2865123, "ms_per_call_p10": 0.27, "ms_per_call_p90": 0.31, "calls": 200, "cells": 31234,
                                "grid": [400, 3000], "blocks": 104, "components": [45, 48], "geometry_bound": True, "init_func_s": 4.2, "finite": True,
                                "cpu_baseline": {"value": 812.3456, "unit": "ms per grid solve", "cores": 8, "kind": "port", "sample": "y" * 300}})
    line = b.compact_line(detail)
    assert len(line) < LINE_LIMIT
    d = json.loads(line)
    assert d["value_p50"] == 29123.5 and d["value_p10"] == 27001.2 and d["value_p90"] == 30555.9 and d["frac_pass"] == 0.1612
    assert d["shipped_case"] == {"ms_per_call": 0.2865, "p10": 0.27, "p90": 0.31, "cells": 31234, "grid": [400, 3000], "blocks": 104, "cpu_ms": 812.3}
    assert d["value_p10"] <= d["value_p50"] <= d["value_p90"]
    # a failed leg is reported, not fatal, and bounded
    line = b.compact_line(dict(detail, shipped_case={"error": "z" * 1000}))
    assert len(line) < LINE_LIMIT and len(json.loads(line)["shipped_case"]["error"]) <= 80


def test_eight_ranks_dry_run(tmp_path):
    """First contact of the N = 8 launch, rehearsed on the CPU: `bench.py --gpus 8 --dry-run` starts eight gloo ranks itself,
    the process group reports 8, configs[3]'s 64 cases are sharded 8 per rank (rank 3 owns cases 24..31) and the N > 1 line
    obeys the same length bound."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSM_BENCH_BACKEND="gloo", PSM_BENCH_LOGDIR=str(tmp_path), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout) < LINE_LIMIT
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world_size_reported"] == 8 and d["dry_run"] is True and len(d["devices"]) == 8
    cb = d["case_batch"]
    assert cb["total_cases"] == 64 and cb["cases_per_step_per_gpu"] == 8 and cb["shards"][3] == [24, 8]
    assert [s[0] for s in cb["shards"]] == list(range(0, 64, 8)) and sum(s[1] for s in cb["shards"]) == 64
    assert (tmp_path / "bench_detail.json").exists() and all((tmp_path / f"bench_rank{r}.err").exists() for r in range(8))


def test_dry_run_under_torch_distributed_run(tmp_path):
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- the environment's ranks are used (no self-launch), rank 0 prints the one line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSM_BENCH_BACKEND="gloo", PSM_BENCH_LOGDIR=str(tmp_path), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_reported"] == 2 and d["case_batch"]["shards"] == [[0, 8], [8, 8]]


def test_launcher_stops_its_ranks_when_it_is_terminated(tmp_path):
    """SIGTERM to the launcher (an outer `timeout`): its ranks are stopped with it, none is left parked in a barrier."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSM_BENCH_BACKEND="gloo", PSM_BENCH_LOGDIR=str(tmp_path), PSM_BENCH_DRY_SLEEP="60")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    pidfile = tmp_path / "rank_pids"
    t_end = time.time() + 60
    while time.time() < t_end and not (pidfile.exists() and len(pidfile.read_text().split()) == 2):
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    _, err = p.communicate(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM and "the ranks were stopped" in err
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid


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
    d, full = run_bench()
    # ---- the line the driver parses
    assert KEYS <= set(d) and ROOF <= set(d["roofline"])
    assert d["metric"].startswith("pressure-solves/sec") and d["unit"] == "solves/s" and d["n_gpus"] == 1
    assert d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 1000 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["geometry"].startswith("bound once")
    assert "device-resident" in d["config"]["value_is"] and d["config"]["degenerate"].startswith("build-defined skip")
    assert "x6" in d["config"]["arithmetic"] and d["config"]["guard_trips"] == 0
    assert all(len(v) <= 160 for v in d["config"].values() if isinstance(v, str))
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["avg_launch_us"] > 0 and abs(r["algorithmic_bytes"] / (r["avg_launch_us"] * 1e-6) / 1e9 - r["achieved"]) < 1e-3 * r["achieved"]
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0 and len(cb["sample"]) <= 120
    assert d["l2_vs_oracle"] < 1e-5 and d["value_end_to_end"] > 1000 and len(d["devices"]) == 1
    gp = d["l2_vs_reference_goldens"]                     # outputs of the reference's own statements, one golden per block layout
    assert set(gp) == {"gradp_272x288", "deltas_256x256", "chapter5_300x400"} and all(0 < v < 5e-5 for v in gp.values()), gp
    # round 5: per-solve quantiles, the headline's whole-solve fraction, the reference's deployment shape in ms per call
    assert d["value_p10"] <= d["value_p50"] <= d["value_p90"] and 0.5 * d["value"] < d["value_p50"] < 1.5 * d["value"]
    assert 0 < d["frac_pass"] < 1 and abs(d["frac_pass"] - full["roofline"]["whole_solve"]["frac"]) < 1e-3
    sc = d["shipped_case"]
    assert sc["grid"] == [400, 3000] and sc["blocks"] == 104 and 20000 < sc["cells"] < 250000 and 0.02 < sc["ms_per_call"] < 5 and sc["p10"] <= sc["ms_per_call"] <= sc["p90"], sc
    assert sc["cpu_ms"] > sc["ms_per_call"] and full["shipped_case"]["finite"] is True and full["shipped_case"]["components"] == [45, 48]
    assert full["per_solve_quantiles"]["samples"] >= 200 and full["per_solve_quantiles"]["solves"] >= 10000
    cbat = d["case_batch"]
    assert cbat["cases_per_step_per_gpu"] == 8 and cbat["value"] > 1000 and cbat["total_cases"] == 8 and cbat["guard_trips"] == 0
    assert cbat["bound"] in ("mfma", "hbm") and 0 < cbat["frac"] < 1 and cbat["l2_vs_oracle"] < 1e-5
    cs = full["case_streams"]                                 # four independent batch-1 streams on the card (never the headline)
    assert cs["streams"] == 4 and cs["finite"] is True and cs["guard_trips"] == 0 and d["case_streams4"] > 0.8 * d["value"], cs
    assert set(d["legs"]) == {"config2", "config4", "unet", "unet8", "unet8_bf16", "unet512_bf16", "unet64_bf16"}
    assert full["legs"]["unet64_bf16"]["cases_per_step_per_gpu"] == 64 and full["legs"]["config4"]["roofline"]["frac"] > 0.05   # (config4: a byte-carrying launch, not a 0.26 MB layer)
    for name, leg in d["legs"].items():
        assert leg["ms_per_step"] > 0 and leg["value"] > 100 and leg["l2_vs_oracle"] < (2e-2 if leg["dtype"] == "bf16" else 1e-5), name
        assert leg["bound"] in ("mfma", "hbm") and 0 < leg["frac"] < 1 and 0 < leg["frac_pass"] < 1, name
        assert (leg["cpu"] is not None and leg["cpu"] > 0) or name.startswith("config"), name
    # ---- the full record (bench_detail.json)
    assert full["value"] == d["value"] and full["ms_per_step"] == d["ms_per_step"]
    fr = full["roofline"]
    top = max(fr["kernels"], key=lambda k: k["avg_us"] * k["launches_per_solve"])
    assert fr["kernel"] == top["name"] == r["kernel"] and len(fr["kernels"]) >= 5 and all(k["launches_per_solve"] == 1 for k in fr["kernels"])
    assert all(k["peak_TFLOPs"] in (157.3, 2516.6 / 6) for k in fr["kernels"])            # every launch priced against its own pipe
    assert all(k["achieved_TFLOPs"] is None or k["achieved_TFLOPs"] < k["peak_TFLOPs"] for k in fr["kernels"])
    e = full["end_to_end"]
    assert e["matches_device_resident_result"] is True and e["hw_queues"]["hip_initialised_before_it_was_set"] is False
    assert {"sync_pageable", "ring_pageable_depth4", "ring_registered_depth4", "ring_zero_copy_depth4"} <= set(e["solves_per_s_per_rank"])
    ps = e["psm_solve"]                                   # the solver boundary (py_func of PythonComm.H) on registered buffers
    assert ps.get("finite") is True and 10 < ps["psm_solve_us"] < 500 and ps["cells"] > 10000, ps
    fc = full["case_batch"]
    assert "configs[3]" in fc["workload"] and fc["gathered_shape"] == [8, 256, 256, 1]
    x6 = [k for k in fc["roofline"]["kernels"] if "x6" in k["name"]]
    assert x6 and all(abs(k["peak_TFLOPs"] - 2516.6 / 6) < 1e-6 and k["achieved_TFLOPs"] < k["peak_TFLOPs"] for k in x6)     # x6 launch: bf16 peak / 6
    for name, leg in full["legs"].items():
        assert ROOF <= set(leg["roofline"]), name
        if name.startswith("unet"):
            wp = leg["roofline"]["whole_pass"]
            assert 0 < wp["frac"] < 1 and all(l["achieved_TFLOPs"] < l["peak_TFLOPs"] for l in leg["roofline"]["launches"]), name
    assert full["legs"]["unet512_bf16"]["grid"] == [512, 512] and full["legs"]["unet512_bf16"]["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_general_path_and_conv_bench_lines():
    d, _ = run_bench("--no-bind", "--no-cpu-baseline")
    assert d["config"]["geometry"].startswith("general path") and "cpu_baseline" not in d
    u, full = run_bench("--workload", "unet", "--no-cpu-baseline")
    assert KEYS <= set(u) and u["roofline"]["bound"] in ("mfma", "hbm") and u["value"] > 100 and "arithmetic" in u["config"]
    assert full["roofline"]["whole_pass"]["achieved_TFLOPs"] > 1
