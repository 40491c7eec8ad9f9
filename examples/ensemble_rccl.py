"""BASELINE config 3 as a user would run it: N random-obstacle 256x256 cases sharded over the GPUs of one node
(one process per GPU, RCCL), every rank solving its contiguous shard with the PCA surrogate; the result shards
stay on their GPUs unless --gather asks for the whole batch on every rank (one all-gather, 2.1 MB per rank at 8
cases).  Rank 0 builds (or reads) the model once and broadcasts it.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/ensemble_rccl.py --cases 64 --gather
  python examples/ensemble_rccl.py --cases 8          # single GPU
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psm_amd  # noqa: E402
from psm_amd import dist as pdist, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=64)
    ap.add_argument("--gather", action="store_true")
    args = ap.parse_args()
    rank, world, local_rank = pdist.env_world()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    pdist.init("nccl", dev)
    model = pdist.broadcast_model(synthetic.make_model("deltas") if rank == 0 else None, device=dev)
    first, count = pdist.shard_cases(args.cases, world, rank)
    grids = synthetic.random_obstacle_cases(args.cases, 256, 256, seed=3).astype(np.float32)[first:first + count]
    d_in = torch.from_numpy(grids).to(dev)
    d_out = torch.empty((count, 256, 256, model.c_out), dtype=torch.float32, device=dev)
    with psm_amd.GridSurrogate(model, 256, 256, max_cases=max(count, 1), device=local_rank) as sur:
        if count:
            sur.solve_device(d_in.data_ptr(), count, d_out.data_ptr(), 0)
        sur.synchronize()
    if args.gather:
        full = pdist.gather_cases(d_out, args.cases)
        if rank == 0:
            print(f"gathered {tuple(full.shape)} on every rank; case means {full.mean(dim=(1, 2, 3))[:4].tolist()} ...")
    else:
        print(f"rank {rank}: cases [{first}, {first + count}) solved, mean {float(d_out.mean()) if count else 0.0:.6f}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
