#!/usr/bin/env python3
"""bench.py -- pressure-solves/sec of the surrogate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling: the per-GPU work is fixed): BASELINE.json
configs[1] -- 256x256 channel-with-obstacle U_to_gradP inference, batch 1, fp32,
P_i = P_o = 128, MLP 3x512 (SURVEY.md §8 d config 1; synthetic seeded input,
seeded random-init weights of that architecture).  One step = one solve
(grid[256,256,3] -> fields[256,256,2]) through the C-ABI with the input already
resident in HBM; steps are issued back to back on one stream (plain launches),
K steps are timed between barrier + synchronize on both sides, MAX over ranks.
For N > 1 each rank drives its own GPU with its own independent cases (the case
batch is sharded, no data-path collective); RCCL is used only for the barrier
and the max-reduction of the time.

Extra objects on the JSON line:
  roofline     dominant kernel (encode) : algorithmic bytes per launch / average
               launch duration measured with HIP events on the launch stream
               (hipExtLaunchKernel start/stop events = the dispatch's own begin/end) in
               an instrumented pass over the same K steps right after the timed region;
               "traffic" = HBM-side bytes per launch from the committed PMC run
               (profiles/pmc_encode.json: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction).
  cpu_baseline the NumPy oracle ("port" of the reference's algorithm, float64 PCA +
               float32 MLP like the reference) timed on the host cores, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 128
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

# BASELINE.json configs as (variant, Ny, Nx, cases per step per GPU, precision, description)
WORKLOADS = {
    "config1": ("gradp", 256, 256, 1, "f32", "BASELINE configs[1]: 256x256 channel+obstacle U_to_gradP, batch 1, fp32, "
                "B=30 blocks of 128x128x3, P_i=P_o=128, MLP 3x512, one independent case stream per GPU"),
    "config2": ("deltas", 256, 256, 1, "f32", "BASELINE configs[2]: 256x256 deltaU_to_deltaP, sequential solves (PISO correctors), "
                "B=9 blocks, P=128, MLP 3x512"),
    "config3": ("deltas", 256, 256, 8, "f32", "BASELINE configs[3]: random-obstacle 256x256 cases, 8 per GPU per step, "
                "B=9 blocks per case, P=128, MLP 3x512"),
    "config4": ("deltas", 512, 512, 1, "bf16", "BASELINE configs[4]: 512x512 high-Re cylinder, bf16 operands / f32 accumulate, "
                "B=30 blocks, P=128, MLP 3x512"),
}


def algorithmic_bytes(model, ny, nx, wbytes=4):
    """SURVEY.md §8(d) BYTES formula, split per kernel group (float32 = 4 B)."""
    S2 = model.S ** 2
    enc = 4 * (ny * nx * model.c_in + S2 * model.c_in) + wbytes * S2 * model.c_in * model.p_in
    dec = 4 * (S2 * model.c_out + ny * nx * model.c_out) + wbytes * S2 * model.c_out * model.p_out
    mlp = wbytes * sum(W.size for W, b in model.weights) + 4 * sum(b.size for W, b in model.weights)
    return {"encode": enc, "decode": dec, "mlp": mlp, "total": enc + dec + mlp}


def cpu_baseline(model, grid, precision="f32", budget_s=12.0, max_solves=2000):
    """Time the oracle on the host cores (bounded sample of the same workload)."""
    from oracle import psm_oracle as orc
    from psm_amd import hostinfo
    cores = hostinfo.available_cpus()            # CPU share of this process (affinity / cgroup quota)
    _limit = hostinfo.limit_blas_threads(cores)  # BLAS pool = the threads actually used
    sc = orc.Scaler(model.scaler_kind, model.in_a, model.in_b, model.out_a, model.out_b)
    om = orc.Model(model.variant, model.c_in, model.c_out, model.comp_in, model.mean_in, model.comp_out,
                   model.mean_out, model.weights, sc, model.out_scale, model.S, model.ov, model.sdf_ch)
    g = grid.astype(np.float64)
    orc.solve_grid(g, om, precision=precision)  # warm-up (BLAS threads, page faults)
    n, t0 = 0, time.perf_counter()
    while n < max_solves and (time.perf_counter() - t0) < budget_s:
        sol = orc.solve_grid(g, om, precision=precision)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} sequential {grid.shape[0]}x{grid.shape[1]} {model.variant} solves of the NumPy oracle "
                      f"(float64 PCA/reassembly, float32 MLP) in {dt:.1f} s"}, sol


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)

UNET_WORKLOADS = {           # the convolutional path (SURVEY.md section 8 row a-conv, parity unpinned) -- not the headline
    "unet": (256, 256, 1, "UNet-S (build-defined: 3x3 convs x2 per level, widths 16-32-64-128-256, max-pool, nearest "
             "upsample + skip concat, 1x1 head), 256x256x3 -> 256x256x1, batch 1, fp32, 7.0 GFLOP per solve"),
    "unet8": (256, 256, 8, "UNet-S, 256x256x3 -> 256x256x1, 8 cases per step per GPU, fp32"),
    "unet8_bf16": (256, 256, 8, "UNet-S, 256x256x3 -> 256x256x1, 8 cases per step per GPU, bf16 operands / f32 accumulate"),
    "unet512_bf16": (512, 512, 1, "UNet-S, 512x512x3 -> 512x512x1 (BASELINE configs[4] shape), batch 1, bf16 operands / f32 accumulate"),
}
MFMA_BF16_PEAK_TFLOPS = 2516.6   # dense bf16 matrix peak (16x the f32 rate)


def main_unet(args):
    """Same protocol for the convolutional path: K forward passes back to back, input resident in HBM."""
    import torch
    from psm_amd import UNetSurrogate, dist as pdist, synthetic
    rank, world, local_rank = pdist.env_world()
    backend = os.environ.get("PSM_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("PSM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    pdist.init(backend, torch.device("cuda", local_rank))
    NY, NX, NC, desc = UNET_WORKLOADS[args.workload]
    W = synthetic.unet_he_weights(seed=7)
    prec = "bf16" if args.workload.endswith("bf16") else "f32"
    peak = MFMA_BF16_PEAK_TFLOPS if prec == "bf16" else MFMA_F32_PEAK_TFLOPS
    net = UNetSurrogate(W, NY, NX, max_cases=NC, device=local_rank, precision=prec)
    grids = [np.stack([synthetic.channel_grid(NY, NX, seed=1 + 1000 * rank + 10 * i + k).astype(np.float32) for k in range(NC)])
             for i in range(args.inputs)]
    d_in = [torch.from_numpy(g).cuda() for g in grids]
    d_out = [torch.empty((NC, NY, NX, 1), dtype=torch.float32, device="cuda") for _ in grids]
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        k = i % len(d_in)
        net.forward_device(d_in[k].data_ptr(), NC, d_out[k].data_ptr(), stream)
    dt_max = pdist.timed_region(step, args.steps, args.warmup, torch.cuda.synchronize, "cuda" if backend == "nccl" else "cpu")
    flops = net.flops * NC
    achieved = flops * args.steps / dt_max / 1e12
    ms, _ = net.profile(d_in[0].data_ptr(), NC, d_out[0].data_ptr())
    out = {"metric": "pressure-solves/sec (256x256 U->p inference)", "value": pdist.aggregate_throughput(NC, args.steps, world, dt_max),
           "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": prec, "data": "synthetic",
           "config": {"workload": desc, "grid": [NY, NX], "cases_per_step_per_gpu": NC,
                      "parallelism": f"case-sharded x{world} (no data-path collective)", "parity": "unpinned (no reference network)"},
           "roofline": {"kernel": "psm_conv3x3_kernel (all 18 layers + head, whole forward pass)", "bound": "mfma", "achieved": achieved,
                        "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                        "algorithmic_flops": flops, "per_layer_ms": [float(v) for v in ms]}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import unet_oracle as uo           # cpu_baseline leg only
        from psm_amd import hostinfo
        cores = hostinfo.available_cpus()
        hostinfo.limit_blas_threads(cores)
        g0 = grids[0][0]
        ref = uo.unet_forward(g0, W, precision=prec)
        n, t0 = 0, time.perf_counter()
        while n < 200 and time.perf_counter() - t0 < 12.0:
            uo.unet_forward(g0, W, precision=prec); n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
                               "sample": f"{n} UNet-S forward passes of the NumPy oracle (float64 accumulation) in {dt:.1f} s"}
        got = d_out[0][0].cpu().numpy()
        out["l2_vs_oracle"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    if rank == 0:
        print(json.dumps(out))
    net.close()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inputs", type=int, default=4, help="distinct input grids rotated through (all resident in HBM)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the geometry: general 8-launch path")
    ap.add_argument("--workload", default="config1", choices=sorted(WORKLOADS) + sorted(UNET_WORKLOADS),
                    help="BASELINE.json config to run (default: configs[1], the one the metric is quoted on)")
    args = ap.parse_args()
    if args.workload in UNET_WORKLOADS:
        return main_unet(args)

    import torch
    import psm_amd
    from psm_amd import dist as pdist
    rank, world, local_rank = pdist.env_world()
    # Rehearsal switches (not used by the driver): several ranks on ONE card need the gloo backend and a
    # forced device index, e.g. PSM_BENCH_BACKEND=gloo PSM_BENCH_DEVICE=0 torchrun --nproc-per-node 2 bench.py
    backend = os.environ.get("PSM_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("PSM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    pdist.init(backend, torch.device("cuda", local_rank))
    red_dev = "cuda" if backend == "nccl" else "cpu"

    import psm_amd
    from psm_amd import synthetic
    variant, NY, NX, NC, precision, wl_desc = WORKLOADS[args.workload]
    model = synthetic.make_model(variant, p_in=P, p_out=P)
    sur = psm_amd.GridSurrogate(model, NY, NX, max_cases=NC, device=local_rank, precision=precision)
    # independent cases per rank (different seeds), all resident in HBM before timing
    if NC == 1:
        grids = [synthetic.channel_grid(NY, NX, seed=1 + 1000 * rank + i, noise=0.05 if NY > 256 else 0.02).astype(np.float32)[None]
                 for i in range(args.inputs)]
    else:
        # an ensemble of NC geometries per rank, each advancing in time: the rotated inputs are the same cases with other
        # velocity fields (the SDF channel -- the geometry -- stays)
        base = synthetic.random_obstacle_cases(NC, NY, NX, seed=3 + 1000 * rank).astype(np.float32)
        grids = []
        for i in range(args.inputs):
            g = base.copy()
            g[..., :model.sdf_ch] *= np.float32(1.0 + 0.05 * i)
            grids.append(g)
    d_in = [torch.from_numpy(g).cuda() for g in grids]
    d_out = [torch.empty((NC, NY, NX, model.c_out), dtype=torch.float32, device="cuda") for _ in grids]
    stream = torch.cuda.current_stream().cuda_stream

    # One case stream per GPU = one simulation: its geometry (the flow-cell pattern of the SDF channel) is bound once,
    # outside the timed region, like the reference's computeOnlyOnce / init_func; the rotated inputs differ in the
    # velocity channels only (checked).  --no-bind times the general path, which takes any geometry per call.
    bound = False
    if not args.no_bind:
        masks = [g[..., model.sdf_ch] != 0 for g in grids]
        if all(np.array_equal(masks[0], m) for m in masks[1:]):
            bound = sur.bind_geometry(d_in[0].data_ptr(), on_device=True, n_cases=NC)

    def step(i):
        k = i % len(d_in)
        sur.solve_device(d_in[k].data_ptr(), NC, d_out[k].data_ptr(), stream)

    dt_max = pdist.timed_region(step, args.steps, args.warmup, torch.cuda.synchronize, red_dev)

    # ---- roofline of the dominant kernel: instrumented pass over the same K steps
    ab = algorithmic_bytes(model, NY, NX, 2 if precision == "bf16" else 4)
    profs = [sur.profile(d_in[0].data_ptr(), NC, d_out[0].data_ptr()) for _ in range(5)]     # event-separated groups: best of 5
    prof = {k: min(p[k] for p in profs) for k in profs[0]}
    dom = "encode"
    REPEAT = 1            # every launch of the timed region's pipeline, one event pair each (hipExtLaunchKernel)
    sur.enable_kernel_timing(dom, True, REPEAT)
    for i in range(args.steps):
        step(i)
    tot_ms, launches = sur.kernel_timing(dom)
    sur.enable_kernel_timing(dom, False)
    avg_s = tot_ms / max(launches, 1) * 1e-3
    achieved = ab[dom] / avg_s / 1e9
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_encode.json")
    if os.path.exists(pmc_file) and args.workload == "config1":
        try:
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "psm_encode_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes": ab[dom], "avg_launch_us": avg_s * 1e6, "launches": launches,
                "launches_per_event_pair": REPEAT,
                "per_kernel_ms_one_solve": prof}

    out = {
        "metric": "pressure-solves/sec (256x256 U->p inference)",
        "value": pdist.aggregate_throughput(NC, args.steps, world, dt_max),
        "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": wl_desc,
                   "grid": [NY, NX], "blocks": sur.B, "p_in": P, "p_out": P, "cases_per_step_per_gpu": NC,
                   "parallelism": f"case-sharded x{world} (no data-path collective)",
                   "geometry": ("bound once per case stream (psm_bind_geometry = the reference's computeOnlyOnce / init_func split): "
                                "6 launches per solve, 7 for case batches") if bound else "general path (any geometry per call): 8 launches per solve (9 for case batches)"},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, sol = cpu_baseline(model, grids[0][0], precision)
        out["cpu_baseline"] = cb
        got = d_out[0][0].cpu().numpy()
        ref = sol.fields
        out["l2_vs_oracle"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        out["gpu_over_cpu"] = out["value"] / cb["value"]
    if rank == 0:
        print(json.dumps(out))
    sur.close()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
