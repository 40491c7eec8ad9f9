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
