"""Multi-GPU plumbing of the case-sharded path.

Cases (independent geometries / time steps) are partitioned contiguously over the
ranks, one process per GPU; the data path has no collective.  ``torch.distributed``
(backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used for the barrier
around the timed region and the MAX-reduction of the elapsed time, and for the
two OPTIONAL exchanges §8e names: ``broadcast_model`` (rank 0 reads the artefacts
once and broadcasts them, ~36 MB, at start-up) and ``gather_cases`` (one
all-gather of the per-rank result shards when a single consumer wants the whole
batch).  Neither is on the timed path of ``bench.py``.
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
    # PSM_DIST_FORCE=1: initialise the process group for a single rank too (rehearsal of the RCCL path on a one-GPU box)
    if (world > 1 or (os.environ.get("PSM_DIST_FORCE") and "MASTER_ADDR" in os.environ)) and not dist.is_initialized():
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


# ---------------------------------------------------------------------------------------------------
# optional exchanges: artefact broadcast at start-up, result all-gather
# ---------------------------------------------------------------------------------------------------
def _dist_ready() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_arrays(arrays, src: int = 0, device="cpu"):
    """``arrays`` (dict name -> ndarray) is given on rank ``src`` and ignored elsewhere; every rank returns the
    dict.  One object broadcast for the manifest (names, shapes, dtypes), one byte broadcast for the payload
    (a single large message: xGMI links are per-peer, many small broadcasts would be latency-bound)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not _dist_ready():
        if arrays is None:
            raise ValueError("no process group: the arrays must be given")
        return dict(arrays)
    rank = dist.get_rank()
    manifest = [None]
    if rank == src:
        if arrays is None:
            raise ValueError("the source rank must pass the arrays")
        arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        manifest[0] = [(k, v.shape, v.dtype.str) for k, v in arrays.items()]
    dist.broadcast_object_list(manifest, src=src)
    total = sum(int(np.prod(sh, dtype=np.int64)) * np.dtype(dt).itemsize for _, sh, dt in manifest[0])
    buf = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    if rank == src and total:
        flat = np.concatenate([v.reshape(-1).view(np.uint8) for v in arrays.values()])
        staged = torch.from_numpy(flat)
        buf[:total].copy_(staged.pin_memory() if buf.is_cuda else staged)      # device copies from / to pinned host tensors only
    dist.broadcast(buf, src=src)
    if buf.is_cuda:
        host = torch.empty(buf.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(buf)
        raw = host.numpy()
    else:
        raw = buf.numpy()
    out, off = {}, 0
    for k, sh, dt in manifest[0]:
        n = int(np.prod(sh, dtype=np.int64)) * np.dtype(dt).itemsize
        out[k] = raw[off:off + n].view(np.dtype(dt)).reshape(sh).copy()
        off += n
    return out


def broadcast_model(model, src: int = 0, device="cpu"):
    """A ``synthetic.SurrogateModel`` (or None off the source rank) -> the same model on every rank."""
    import dataclasses
    import numpy as np
    import torch.distributed as dist
    from .synthetic import SurrogateModel
    if not _dist_ready():
        return model
    payload, meta = None, [None]
    if dist.get_rank() == src:
        payload, scal = {}, {}
        for f in dataclasses.fields(model):
            v = getattr(model, f.name)
            if f.name == "weights":
                for i, (W, b) in enumerate(v):
                    payload[f"W{i}"], payload[f"b{i}"] = np.asarray(W), np.asarray(b)
                scal["n_layers"] = len(v)
            elif f.name == "conv1d":
                for i, (K, b) in enumerate(v):
                    payload[f"cK{i}"], payload[f"cb{i}"] = np.asarray(K), np.asarray(b)
                scal["n_conv1d"] = len(v)
            elif isinstance(v, np.ndarray):
                payload["f:" + f.name] = v
            else:
                scal[f.name] = v
        meta[0] = scal
    dist.broadcast_object_list(meta, src=src)
    arrs = broadcast_arrays(payload, src, device)
    scal = dict(meta[0])
    n_layers = scal.pop("n_layers")
    n_conv1d = scal.pop("n_conv1d", 0)
    kw = dict(scal)
    kw.update({k[2:]: v for k, v in arrs.items() if k.startswith("f:")})
    kw["weights"] = [(arrs[f"W{i}"], arrs[f"b{i}"]) for i in range(n_layers)]
    kw["conv1d"] = [(arrs[f"cK{i}"], arrs[f"cb{i}"]) for i in range(n_conv1d)]
    return SurrogateModel(**kw)


def gather_cases(local, n_cases: int):
    """Per-rank result shard ``local`` [count_r, ...] (torch tensor on the rank's device: CUDA under RCCL, CPU under
    gloo; ``count_r`` from ``shard_cases``) -> the whole batch [n_cases, ...] on every rank, in case order.  Uneven
    shards are padded to the largest for the collective (``all_gather_into_tensor`` needs equal sizes)."""
    import torch
    import torch.distributed as dist
    if not _dist_ready():
        if local.shape[0] != n_cases:
            raise ValueError("no process group: the local shard must be the whole batch")
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    first, count = shard_cases(n_cases, world, rank)
    if local.shape[0] != count:
        raise ValueError(f"rank {rank} holds {local.shape[0]} cases, its shard has {count}")
    cmax = (n_cases + world - 1) // world
    send = local.contiguous()
    if count < cmax:
        send = torch.cat([send, send.new_zeros((cmax - count,) + tuple(send.shape[1:]))])
    recv = send.new_empty((world * cmax,) + tuple(send.shape[1:]))
    dist.all_gather_into_tensor(recv, send)
    parts = [recv[r * cmax:r * cmax + shard_cases(n_cases, world, r)[1]] for r in range(world)]
    return torch.cat(parts)
