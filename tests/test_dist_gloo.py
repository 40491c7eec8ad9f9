"""N > 1 path on CPU: two gloo ranks run the bench's timed-region protocol (barrier,
K steps, MAX over ranks, whole-job aggregate) and the contiguous case sharding."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from psm_amd import dist as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_cases_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_cases(n, world, r) for r in range(world)]
            covered = [i for f, c in spans for i in range(f, f + c)]
            assert covered == list(range(n))
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert pdist.shard_cases(64, 8, 3) == (24, 8)          # BASELINE config 3: 8 cases per GPU
    with pytest.raises(ValueError):
        pdist.shard_cases(4, 2, 2)


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, {root!r})
    import torch
    from psm_amd import dist as pdist
    rank, world = pdist.init("gloo")
    assert world == 2
    first, count = pdist.shard_cases(9, world, rank)
    done = []
    def step(i):                      # rank 1 is slower: the MAX over ranks must pick it up
        time.sleep(0.002 * (1 + rank))
        done.append(i)
    dt = pdist.timed_region(step, steps=20, warmup=3)
    total = torch.tensor([count], dtype=torch.int64)
    torch.distributed.all_reduce(total)
    print(json.dumps(dict(rank=rank, first=first, count=count, dt=dt, steps=len(done), total=int(total.item()),
                          value=pdist.aggregate_throughput(1, 20, world, dt))))
    torch.distributed.destroy_process_group()
""")


def test_two_gloo_ranks_timed_region(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [(d["first"], d["count"]) for d in outs] == [(0, 5), (5, 4)]
    assert all(d["total"] == 9 and d["steps"] == 23 for d in outs)
    assert outs[0]["dt"] == outs[1]["dt"]                 # MAX over ranks, identical everywhere
    assert outs[0]["dt"] >= 20 * 0.004 * 0.9              # the slower rank's 20 x 4 ms
    assert abs(outs[0]["value"] - 2 * 20 / outs[0]["dt"]) < 1e-9


WORKER2 = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch
    from psm_amd import dist as pdist, synthetic
    rank, world = pdist.init("gloo")
    # artefacts exist on rank 0 only (as if read from disk there); everyone ends up with the same model
    model = synthetic.make_model("deltas", p_in=8, p_out=8, S=16) if rank == 0 else None
    model = pdist.broadcast_model(model)
    ref = synthetic.make_model("deltas", p_in=8, p_out=8, S=16)
    same = (model.variant == ref.variant and model.scaler_kind == ref.scaler_kind and model.ov == ref.ov
            and np.array_equal(model.comp_in, ref.comp_in) and np.array_equal(model.mean_out, ref.mean_out)
            and all(np.array_equal(a, c) and np.array_equal(b, d) for (a, b), (c, d) in zip(model.weights, ref.weights))
            and np.array_equal(np.asarray(model.in_a), np.asarray(ref.in_a)) and model.comp_in.dtype == ref.comp_in.dtype)
    # 5 cases over 2 ranks (3 + 2): each rank "solves" its shard, the all-gather restores case order
    first, count = pdist.shard_cases(5, world, rank)
    local = torch.stack([torch.full((4, 6, 1), float(first + i)) for i in range(count)])
    full = pdist.gather_cases(local, 5)
    order = [float(full[i, 0, 0, 0]) for i in range(5)]
    print(json.dumps(dict(rank=rank, same=bool(same), order=order, shape=list(full.shape))))
    torch.distributed.destroy_process_group()
""")


def test_two_gloo_ranks_broadcast_model_and_gather(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        d = __import__("json").loads(o.strip().splitlines()[-1])
        assert d["same"] and d["order"] == [0.0, 1.0, 2.0, 3.0, 4.0] and d["shape"] == [5, 4, 6, 1]


def test_exchanges_without_process_group():
    import numpy as np
    import torch
    a = {"x": np.arange(6, dtype=np.float32).reshape(2, 3)}
    out = pdist.broadcast_arrays(a)
    assert np.array_equal(out["x"], a["x"])
    t = torch.zeros(3, 2, 2, 1)
    assert pdist.gather_cases(t, 3) is t
    with pytest.raises(ValueError):
        pdist.gather_cases(t, 4)
