"""Multi-GPU plumbing of the case-sharded path.

Cases (independent geometries / time steps) are partitioned contiguously over the
ranks, one process per GPU; the data path has no collective.  ``torch.distributed``
(backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the
barrier around the timed region and the MAX-reduction of the elapsed time.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Tuple


def shard_cases(n_cases: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced partition: -> (first, count) of this rank's cases."""
    if world < 1 or not (0 <= rank < world) or n_cases < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_cases, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def env_world() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device=None):
    import torch.distributed as dist
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def barrier(sync: Callable[[], None] = lambda: None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    sync()


def max_over_ranks(value: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_region(step: Callable[[int], None], steps: int, warmup: int, sync: Callable[[], None] = lambda: None,
                 device="cpu") -> float:
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    for i in range(warmup):
        step(i)
    barrier(sync)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    barrier(sync)
    return max_over_ranks(dt, device)


def aggregate_throughput(units_per_rank_per_step: int, steps: int, world: int, dt_max: float) -> float:
    """Whole-job units per second: every rank processed the same number of units (weak scaling)."""
    return world * units_per_rank_per_step * steps / dt_max
