#!/usr/bin/env python3
"""bench.py -- pressure-solves/sec of the surrogate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

``--gpus N`` with no torchrun environment starts the N ranks itself (fresh child processes, one per GPU, before this
process touches the GPU); under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` the
environment's ranks are used.  The world size reported by the process group must equal ``--gpus``.

Workload of ``value`` at every N (weak scaling: the per-GPU work is fixed): BASELINE.json configs[1] -- 256x256
channel-with-obstacle U_to_gradP inference, batch 1, fp32, P_i = P_o = 128, MLP 3x512 (SURVEY.md §8 d config 1;
synthetic seeded input, seeded random-init weights of that architecture).  One step = one solve
(grid[256,256,3] -> fields[256,256,2]) through the C-ABI with the input already resident in HBM; steps are issued
back to back on one stream, K steps are timed between barrier + synchronize on both sides, MAX over ranks.  Each rank
drives its own GPU with its own independent case stream (no data-path collective); RCCL carries the model broadcast
(outside the timed region), the barrier and the max-reduction of the time.

Extra objects on the JSON line:
  value_end_to_end / end_to_end   SURVEY §8(d)'s solve INCLUDING the H2D copy of the grid and the D2H copy of the field
               (host buffers in, host buffers out), measured by a C++ loop inside the library through the public C-ABI
               (psm_bench_host): the pinned ring on caller-registered memory, next to the synchronous psm_solve_grid,
               the ring on pageable memory and the zero-copy slot form.
  case_batch   BASELINE configs[3]: random-obstacle 256x256 deltaU_to_deltaP cases, 8 per GPU per step (64 over 8 GPUs),
               one geometry per case slot, aggregate solves/s over all ranks + one all-gather of the result shards
               (outside the timed region).
  roofline     the kernel with the largest measured time in this run: every dispatch of K instrumented solves carries
               its own begin / end stamps (hipExtLaunchKernelGGL events, psm_time_kernels); achieved = that kernel's
               algorithmic bytes per launch / its average duration.  "traffic" = HBM-side bytes per launch from the
               committed PMC run of the same kernel (profiles/r02_pmc.json; null when the kernel source has changed
               since that run).
  cpu_baseline the NumPy oracle ("port" of the reference's algorithm, float64 PCA + float32 MLP like the reference)
               timed on the host cores, rank 0, N=1.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The host-buffer ring keeps several tickets in flight on their own streams: give the HIP runtime enough hardware queues
# for them (read when the runtime initialises, i.e. before torch touches the GPU; libpsm_hip.so asks for the same when it
# is loaded first).  Measured: 50 us per end-to-end solve with 8 queues against 35 with 16.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

P = 128
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2516.6   # dense bf16 matrix peak (16x the f32 rate)

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
UNET_WORKLOADS = {           # the convolutional path (SURVEY.md section 8 row a-conv, parity unpinned) -- not the headline
    "unet": (256, 256, 1, "UNet-S (build-defined: 3x3 convs x2 per level, widths 16-32-64-128-256, max-pool, nearest "
             "upsample + skip concat, 1x1 head), 256x256x3 -> 256x256x1, batch 1, fp32, 7.0 GFLOP per solve"),
    "unet8": (256, 256, 8, "UNet-S, 256x256x3 -> 256x256x1, 8 cases per step per GPU, fp32"),
    "unet8_bf16": (256, 256, 8, "UNet-S, 256x256x3 -> 256x256x1, 8 cases per step per GPU, bf16 operands / f32 accumulate"),
    "unet512_bf16": (512, 512, 1, "UNet-S, 512x512x3 -> 512x512x1 (BASELINE configs[4] shape), batch 1, bf16 operands / f32 accumulate"),
}


# ----------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torchrun environment
# ----------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relay rank 0's
    JSON line.  Called before anything in this process has touched the GPU (nothing is ever re-exec'd)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill()                                  # exactly the child this launcher started
            rcs.append(p.wait())
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0


def algorithmic_bytes(model, ny, nx, wbytes=4):
    """SURVEY.md §8(d) BYTES formula, split per kernel group (float32 = 4 B)."""
    S2 = model.S ** 2
    enc = 4 * (ny * nx * model.c_in + S2 * model.c_in) + wbytes * S2 * model.c_in * model.p_in
    dec = 4 * (S2 * model.c_out + ny * nx * model.c_out) + wbytes * S2 * model.c_out * model.p_out
    layers = [wbytes * W.size + 4 * b.size for W, b in model.weights]
    return {"encode": enc, "decode": dec, "mlp": sum(layers), "layers": layers, "total": enc + dec + sum(layers)}


def kernel_algorithmic_bytes(name, ab, launches_per_solve):
    """Algorithmic bytes one launch of kernel `name` must move (None for kernels whose traffic is an artefact of the
    launch structure: slab sums, strip sums, chain)."""
    nl = len(ab["layers"])
    if "encode" in name:
        return ab["encode"]
    if "decode" in name:
        return ab["decode"]
    if "reduce_dense1" in name:
        return ab["layers"][0]
    if "dense" in name:
        # the remaining layers share this kernel: average bytes of the layers it ran
        rest = ab["layers"][1:] if launches_per_solve < nl else ab["layers"]
        return sum(rest) / max(len(rest), 1)
    return None


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("psm_kernels.hip", "psm_bf16.hip", "psm_kernels.h"):
        p = os.path.join(ROOT, "solving-poisson-s-equation-through-dl-for-cfd-apllications_amd", "csrc", f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kernel, workload):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (tools/pmc_summary.py), only while the
    kernel sources are the ones that were profiled."""
    f = os.path.join(ROOT, "profiles", "r02_pmc.json")
    try:
        d = json.load(open(f))
    except Exception:
        return None, None
    if d.get("workload") != workload or d.get("kernel_source_hash") != kernel_source_hash():
        return None, "profiles/r02_pmc.json is from other kernel sources: not used"
    v = d.get("kernels", {}).get(kernel)
    return (v, "profiles/r02_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2*FETCH + WRITE KiB, gfx950 correction)") if v else (None, None)


def cpu_baseline(model, grid, precision="f32", budget_s=12.0, max_solves=4000):
    """Time the CPU restatements on the host cores (bounded sample of the same workload): the C / OpenMP port
    (oracle/psm_cpu.c, "CPU back-end A" of BASELINE.md) is the reported baseline, the NumPy oracle -- the reference-style
    number -- is timed beside it; the fields for l2_vs_oracle come from the NumPy oracle."""
    import numpy as np
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
    while n < max_solves and (time.perf_counter() - t0) < budget_s / 2:
        sol = orc.solve_grid(g, om, precision=precision)
        n += 1
    dt = time.perf_counter() - t0
    numpy_rate = n / dt
    out = {"value": numpy_rate, "unit": "solves/s", "cores": int(cores), "kind": "port",
           "sample": f"{n} sequential {grid.shape[0]}x{grid.shape[1]} {model.variant} solves of the NumPy oracle "
                     f"(float64 PCA/reassembly, float32 MLP) in {dt:.1f} s"}
    if precision == "f32":
        try:
            from oracle import psm_cpu
            cm = psm_cpu.CpuModel(om)
            f = psm_cpu.solve_grid(g, cm, threads=int(cores))
            agree = float(np.abs(f - sol.fields).max() / np.abs(sol.fields).max())
            n, t0 = 0, time.perf_counter()
            while n < max_solves and (time.perf_counter() - t0) < budget_s / 2:
                psm_cpu.solve_grid(g, cm, threads=int(cores))
                n += 1
            dt = time.perf_counter() - t0
            out = {"value": n / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
                   "sample": f"{n} sequential {grid.shape[0]}x{grid.shape[1]} {model.variant} solves of the C / OpenMP port oracle/psm_cpu.c "
                             f"(float64 PCA/reassembly, float32 MLP, {int(cores)} threads) in {dt:.1f} s",
                   "numpy_oracle_solves_per_s": numpy_rate, "max_rel_diff_vs_numpy_oracle": agree}
        except Exception as e:                   # no compiler on the box: the NumPy oracle stays the baseline
            out["c_port"] = f"not available: {e!r}"[:200]
    return out, sol


def degenerate_note(variant, ny, nx, S=128):
    """BASELINE configs whose last block row duplicates the previous one: the reference itself is undefined there."""
    stride = 32 if variant == "gradp" else (96 if variant == "deltas" else None)
    if stride is None or (ny - S) % stride != 0:
        return None
    where = "UGP:340 -> UGP:359 (mean of an empty slice, NaN field)" if variant == "gradp" else "SMD:335 (broadcast error)"
    return ("build-defined skip: p_i == 0 on this grid, where the reference itself is undefined (" + where + "); the duplicate "
            "last block row is encoded / decoded but left out of the reassembly, and l2_vs_oracle compares with the "
            "oracle's same skip mode (the golden vectors use non-degenerate grids)")


def finish(pdist_mod=None):
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


def main_dry(args):
    """Launcher / process-group plumbing only (CPU tests): no GPU work, value is null."""
    from psm_amd import dist as pdist
    import torch.distributed as dist
    rank, world, _ = pdist.env_world()
    pdist.init(os.environ.get("PSM_BENCH_BACKEND", "gloo"))
    if world != args.gpus:
        raise SystemExit(f"world size {world} != --gpus {args.gpus}")
    dt = pdist.timed_region(lambda i: time.sleep(1e-4), args.steps, args.warmup)
    reported = dist.get_world_size() if dist.is_initialized() else 1
    if rank == 0:
        print(json.dumps({"metric": "pressure-solves/sec (256x256 U->p inference)", "value": None, "unit": "solves/s",
                          "n_gpus": world, "world_size_reported": reported, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "dry_run": True, "data": "none (dry run: launcher and "
                          "process-group plumbing only)"}))
    finish()


def main_unet(args):
    """Same protocol for the convolutional path: K forward passes back to back, input resident in HBM."""
    import numpy as np
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
    finish()


def host_rates(sur, grids, n_cases, steps, warmup, modes):
    """psm_bench_host (C++ loop through the public C-ABI): {mode name: (solves/s per rank, last field)}."""
    import ctypes as C
    import numpy as np
    host = np.ascontiguousarray(np.stack(grids))                     # [n_inputs][n_cases, ny, nx, c_in]
    out = {}
    names = {0: "sync_pageable", 1: "ring_pageable", 2: "ring_registered", 3: "ring_zero_copy"}
    for mode, depth in modes:
        last = np.empty((n_cases, sur.ny, sur.nx, sur.model.c_out), np.float32)
        sec = C.c_double()
        sur._chk(sur.lib.psm_bench_host(sur.h, host.ctypes.data_as(C.POINTER(C.c_float)), host.shape[0], n_cases, mode, depth,
                                        steps, warmup, C.byref(sec), last.ctypes.data_as(C.POINTER(C.c_float))))
        out[names[mode] + (f"_depth{depth}" if mode else "")] = (n_cases * steps / sec.value, last)
    return out


def time_kernels(sur, d_grid, n_cases, d_fields, steps):
    """psm_time_kernels -> [(name, avg_us, launches)] in launch order."""
    import ctypes as C
    cap = 32
    names = C.create_string_buffer(cap * 64)
    ms = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    nk = C.c_int32()
    sur._chk(sur.lib.psm_time_kernels(sur.h, C.c_void_p(d_grid), n_cases, C.c_void_p(d_fields), steps, names, ms, cnt, cap, C.byref(nk)))
    out = []
    for k in range(min(nk.value, cap)):
        nm = names.raw[k * 64:(k + 1) * 64].split(b"\0", 1)[0].decode()
        out.append((nm, ms[k] / max(cnt[k], 1) * 1e3, int(cnt[k])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inputs", type=int, default=4, help="distinct input grids rotated through (all resident in HBM)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the geometry: general 8-launch path")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end and case-batch legs")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group plumbing only (no GPU work)")
    ap.add_argument("--workload", default="config1", choices=sorted(WORKLOADS) + sorted(UNET_WORKLOADS),
                    help="BASELINE.json config to run (default: configs[1], the one the metric is quoted on)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # N ranks asked for and no launcher environment: start them ourselves, before this process touches the GPU
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.dry_run:
        return main_dry(args)
    if args.workload in UNET_WORKLOADS:
        return main_unet(args)

    import numpy as np
    import torch
    import psm_amd
    from psm_amd import dist as pdist, synthetic
    rank, world, local_rank = pdist.env_world()
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus} or without a launcher")
    # Rehearsal switches (not used by the driver): several ranks on ONE card need the gloo backend and a
    # forced device index, e.g. PSM_BENCH_BACKEND=gloo PSM_BENCH_DEVICE=0 python bench.py --gpus 2
    backend = os.environ.get("PSM_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("PSM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    pdist.init(backend, torch.device("cuda", local_rank))
    red_dev = "cuda" if backend == "nccl" else "cpu"
    import torch.distributed as dist
    world_reported = dist.get_world_size() if dist.is_initialized() else 1
    if world_reported != args.gpus:
        raise SystemExit(f"process group reports {world_reported} ranks, --gpus {args.gpus}")

    variant, NY, NX, NC, precision, wl_desc = WORKLOADS[args.workload]
    # rank 0 builds the artefacts, every other rank receives them over RCCL (one byte broadcast, outside the timed region)
    t_b = time.perf_counter()
    model = synthetic.make_model(variant, p_in=P, p_out=P) if rank == 0 else None
    model = pdist.broadcast_model(model, 0, red_dev)
    broadcast_s = time.perf_counter() - t_b
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
    torch.cuda.synchronize()
    got_dev = d_out[0].cpu().numpy()

    # ---- roofline of the kernel with the largest measured time: every dispatch of an instrumented pass over the same
    # K steps carries its own begin / end stamps
    ab = algorithmic_bytes(model, NY, NX, 2 if precision == "bf16" else 4)
    kt = time_kernels(sur, d_in[0].data_ptr(), NC, d_out[0].data_ptr(), args.steps)
    per_solve = {nm: n / args.steps for nm, _, n in kt}
    # time per solve of each kernel (a kernel launched several times per solve counts with all its launches)
    dom_name, dom_us, dom_n = max(kt, key=lambda r: r[1] * per_solve[r[0]])
    kernels = []
    for nm, us, n in kt:
        b = kernel_algorithmic_bytes(nm, ab, per_solve[nm])
        kernels.append({"name": nm, "avg_us": us, "launches_per_solve": per_solve[nm], "algorithmic_bytes": b,
                        "achieved_GBs": (b / (us * 1e-6) / 1e9) if b else None})
    dom_bytes = kernel_algorithmic_bytes(dom_name, ab, per_solve[dom_name])
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_bytes else 0.0
    traffic, traffic_src = committed_traffic(dom_name, args.workload) if bound else (None, None)
    roofline = {"kernel": dom_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": dom_bytes, "avg_launch_us": dom_us, "launches": dom_n,
                "selection": "largest measured time per solve among all kernels of the instrumented pass",
                "whole_solve": {"algorithmic_bytes": ab["total"], "achieved_GBs": ab["total"] * NC / (dt_max / args.steps) / 1e9,
                                "frac": ab["total"] * NC / (dt_max / args.steps) / 1e9 / HBM_PEAK_GBS},
                "kernels": kernels}

    out = {
        "metric": "pressure-solves/sec (256x256 U->p inference)",
        "value": pdist.aggregate_throughput(NC, args.steps, world, dt_max),
        "unit": "solves/s", "n_gpus": world, "world_size_reported": world_reported, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": wl_desc,
                   "grid": [NY, NX], "blocks": sur.B, "p_in": P, "p_out": P, "cases_per_step_per_gpu": NC,
                   "parallelism": f"case-sharded x{world} (no data-path collective; model broadcast from rank 0 over "
                                  f"{'RCCL' if backend == 'nccl' else backend} before the timed region: {broadcast_s * 1e3:.0f} ms)",
                   "value_is": "device-resident rate: the input grid is already in HBM when the timed region starts and the "
                               "field stays in HBM (the bench contract); the H2D/D2H-inclusive solve of SURVEY section 8(d) is "
                               "value_end_to_end",
                   "degenerate": degenerate_note(variant, NY, NX),
                   "geometry": ("bound once per case stream (psm_bind_geometry = the reference's computeOnlyOnce / init_func split): "
                                "6 launches per solve, 7 for case batches") if bound else "general path (any geometry per call): 8 launches per solve (9 for case batches)"},
        "roofline": roofline,
    }

    # ---- one solve = host grid in, host field out (H2D + D2H included)
    if not args.no_extras:
        n_e2e = max(200, min(args.steps, 3000))
        modes = [(2, 4)] if world > 1 else [(0, 1), (1, 4), (2, 1), (2, 2), (2, 4), (3, 4)]
        pdist.barrier(torch.cuda.synchronize)
        wu_e2e = min(args.warmup, 100)
        rates = host_rates(sur, grids, NC, n_e2e, wu_e2e, modes)
        key = "ring_registered_depth4"
        last_in = (wu_e2e + n_e2e - 1) % len(grids)                          # input of the last end-to-end solve
        slow = pdist.max_over_ranks(1.0 / rates[key][0], red_dev)            # slowest rank bounds the job
        out["value_end_to_end"] = world * 1.0 / slow
        out["end_to_end"] = {
            "what": "host buffers in, host buffers out: H2D of the grid and D2H of the field included (SURVEY section 8(d)); "
                    "C++ loop inside the library through the public C-ABI (psm_bench_host); ring slots with their own "
                    "stream and scratch: DMA copy in, one hipGraph replay of the kernels, DMA copy out per ticket",
            "value_is": key + " (psm_submit_grid_io / psm_wait_grid on caller-registered memory, 4 tickets in flight)",
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "steps": n_e2e, "solves_per_s_per_rank": {k: v[0] for k, v in rates.items()},
            "matches_device_resident_result": bool(np.array_equal(rates[key][1], d_out[last_in].cpu().numpy()))}

    # ---- BASELINE configs[3]: the case batch, 8 random-obstacle cases per GPU per step
    if not args.no_extras and args.workload == "config1":
        v3, ny3, nx3, nc3, prec3, desc3 = WORKLOADS["config3"]
        m3 = synthetic.make_model(v3, p_in=P, p_out=P) if rank == 0 else None
        m3 = pdist.broadcast_model(m3, 0, red_dev)
        total_cases = nc3 * world
        first, count = pdist.shard_cases(total_cases, world, rank)           # contiguous shard of the case batch
        allc = synthetic.random_obstacle_cases(count, ny3, nx3, seed=3 + 1000 * rank).astype(np.float32)
        sur3 = psm_amd.GridSurrogate(m3, ny3, nx3, max_cases=count, device=local_rank, precision=prec3)
        g3 = []
        for i in range(args.inputs):
            g = allc.copy()
            g[..., :m3.sdf_ch] *= np.float32(1.0 + 0.05 * i)
            g3.append(torch.from_numpy(g).cuda())
        o3 = [torch.empty((count, ny3, nx3, m3.c_out), dtype=torch.float32, device="cuda") for _ in g3]
        b3 = sur3.bind_geometry(g3[0].data_ptr(), on_device=True, n_cases=count) if not args.no_bind else False

        def step3(i):
            k = i % len(g3)
            sur3.solve_device(g3[k].data_ptr(), count, o3[k].data_ptr(), stream)
        k3 = max(100, args.steps // 4)
        dt3 = pdist.timed_region(step3, k3, max(10, args.warmup // 4), torch.cuda.synchronize, red_dev)
        whole = pdist.gather_cases(o3[0] if red_dev == "cuda" else o3[0].cpu(), total_cases)     # one all-gather, untimed
        out["case_batch"] = {"workload": desc3, "value": total_cases * k3 / dt3, "unit": "solves/s", "steps": k3,
                             "ms_per_step": dt3 / k3 * 1e3, "cases_per_step_per_gpu": count, "total_cases": total_cases,
                             "geometry": "one bound geometry per case slot (7 launches per step)" if b3 else "general path (9 launches per step)",
                             "gathered_shape": list(whole.shape)}
        sur3.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, sol = cpu_baseline(model, grids[0][0], precision)
        out["cpu_baseline"] = cb
        ref = sol.fields
        out["l2_vs_oracle"] = float(np.linalg.norm(got_dev[0] - ref) / np.linalg.norm(ref))
        out["gpu_over_cpu"] = out["value"] / cb["value"]
    if rank == 0:
        print(json.dumps(out))
    sur.close()
    finish()


if __name__ == "__main__":
    main()
