#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MPC QP hot path on MI355X.

Metric (BASELINE.json): QP solves/sec at batch 65,536, nx=12 nu=4 N=30 (quadrotor hover MPC,
config 3), fp64, synthetic x0 (numpy default_rng(seed)), random-free model.  One "step" = one
pass of the hot path over one batch: mpcqp_solve (ADMM kernel + polish kernel) with x0 already
resident in HBM; the model set-up (assemble + factor) is outside the timed region because the
model is shared and fixed (LTI) -- it is reported as `setup_ms`.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as torch.distributed.run with one rank per GPU; every rank solves
its own 65,536-instance shard (weak scaling, no data-path collective), the step ends with the
all-reduce of the "all solved" flag the north star names; the all-gather of the solutions is timed
separately (config.allgather_ms) because its volume grows with N and a caller only needs it when
the host wants every trajectory.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0
ADMM_TRAFFIC_BYTES_65536 = (2 * 49314.40 + 213303.90) * 1024.0   # profiles/r01_v5_pmc.csv
POLISH_A_TRAFFIC_BYTES_65536 = (2 * 7200.65 + 185924.27) * 1024.0   # first polish_oct_kernel launch, profiles/r01_v7_pmc.csv


def cpu_baseline(prob, seconds=15.0):
    """Time the C restatement of the reference's solver class (oracle/qp_ipm.c: dense Mehrotra
    IPM on (P,q,G,h) with cvxopt's default tolerances, one QP per thread) on a bounded sample of
    the SAME workload, on this box's host cores.  Checker/baseline only -- never the product."""
    from oracle import cpu_baseline as cb
    from oracle import mpc_oracle as mo
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())
    cores = int(os.environ.get("OMP_NUM_THREADS", cores)) if os.environ.get("OMP_NUM_THREADS", "").isdigit() else cores
    cores = max(1, min(cores, 16))
    # calibrate on 64 instances single-thread, then size the sample to ~`seconds` of wall time
    _, _, _, t1 = cb.solve_batch_box(H, F, f0, lb, ub, prob["x0"][:64], nthreads=1)
    per_core = 64 / t1
    nsamp = int(min(len(prob["x0"]), max(256, per_core * cores * seconds)))
    U, it, bad, t = cb.solve_batch_box(H, F, f0, lb, ub, prob["x0"][:nsamp], nthreads=cores)
    return {"value": nsamp / t, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"first {nsamp} of the 65,536 x0 (same H,F,bounds); dense Mehrotra IPM, cvxopt default "
                      f"tolerances, mean {float(it.mean()):.1f} iterations, {bad} not optimal; "
                      f"single-core {per_core:.0f} solves/s",
            "single_core": per_core}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--rho", type=float, default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL; MPC_BENCH_BACKEND=gloo with several ranks on ONE device is only a rehearsal of
        # this code path on a single-GPU box
        backend = os.environ.get("MPC_BENCH_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    verbose = bool(os.environ.get("MPC_BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    from mpc_amd.configs import make_config
    from mpc_amd.planner import Planner

    B = args.batch
    prob = make_config(3, batch=B, seed=rank)       # every rank draws its own x0 shard
    pl = Planner(prob["nx"], prob["nu"], prob["N"], device=dev)
    pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    pl.setup(rho=args.rho)
    torch.cuda.synchronize(dev)
    setup_ms = (time.perf_counter() - t0) * 1e3
    x0 = torch.as_tensor(prob["x0"], device=dev)
    opts = {}
    if args.max_iter is not None:
        opts["max_iter"] = args.max_iter
    # collectives run on device tensors over RCCL; a rehearsal backend (gloo) gets host tensors
    cdev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
    flag = torch.zeros(1, dtype=torch.int32, device=cdev)
    bar = torch.zeros(1, dtype=torch.int32, device=cdev)

    def barrier():
        # RCCL: the library's barrier; rehearsal backends: an all-reduce, which synchronises the ranks just the same
        if dist.get_backend() == "nccl":
            dist.barrier()
        else:
            dist.all_reduce(bar)

    flags = [flag, flag.clone()]
    pending = [None]
    nstep = [0]

    def step():
        # the all-reduce of the all-solved flag of step i is in flight while step i+1 computes (two flag buffers);
        # the timed region ends with drain() + barrier, so every collective is inside it
        U = pl.solve(x0, **opts)
        if dist is not None:
            f = flags[nstep[0] & 1]
            nstep[0] += 1
            f.copy_((pl.info.status == 0).all().to(torch.int32).reshape(1), non_blocking=True)
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = dist.all_reduce(f, op=dist.ReduceOp.MIN, async_op=True)
        return U

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    note("setup done")
    for _ in range(args.warmup):
        U = step()
    drain()
    note("warmup done")
    if dist is not None:
        barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        U = step()
    drain()
    if dist is not None:
        barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    note("timed region done")
    info = pl.info
    n_solved = int((info.status == 0).sum().item())
    iters_mean = float(info.iters.float().mean().item())
    iters_max = int(info.iters.max().item())
    psteps_mean = float(info.polish_steps.float().mean().item())
    psteps_max = int(info.polish_steps.max().item())

    # ---- per-kernel device time with events on the stream the kernels run on (torch's current stream)
    def timed(fn, reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps       # ms

    reps = max(5, min(args.steps, 20))
    admm_on = iters_max > 0
    admm_ms = timed(lambda: pl.solve(x0, stages=1, **opts), reps) if admm_on else 0.0
    both_ms = timed(lambda: pl.solve(x0, **opts), reps)
    polish_ms = max(both_ms - admm_ms, 0.0)
    # the first (and longest) active-set launch alone: stages bit 2 stops after it (diagnostic; queued instances stay open)
    # (bit 3: after the cold-start ordering kernels, which run before it)
    order_ms = max(timed(lambda: pl.solve(x0, stages=15, **opts), reps) - admm_ms, 0.0)
    pass_a_ms = max(timed(lambda: pl.solve(x0, stages=7, **opts), reps) - admm_ms - order_ms, 0.0)
    # p50 / p99 of single-batch solve latency: one event pair per solve
    lat = []
    nlat = max(20, min(args.steps, 100))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlat)]
    for e0, e1 in evs:
        e0.record(); pl.solve(x0, **opts); e1.record()
    torch.cuda.synchronize(dev)
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    p50_ms, p99_ms = lat[len(lat) // 2], lat[min(len(lat) - 1, int(0.99 * len(lat)))]

    note("per-kernel timing done")
    allgather_ms = None
    if dist is not None and cdev == dev:
        gathered = torch.empty((world,) + tuple(U.shape), dtype=U.dtype, device=dev)
        allgather_ms = timed(lambda: dist.all_gather_into_tensor(gathered, U), 5)

    # ---- roofline of the dominant kernel (the ADMM kernel), SURVEY.md section 8(d) counts:
    # flops/iteration/instance = 2 n^2 + 10 (n + m), m = n box rows, times the iterations executed
    n = pl.n
    flops_per_iter = 2 * n * n + 10 * (n + n)
    admm_flops = B * iters_mean * flops_per_iter
    achieved_tf = admm_flops / (admm_ms * 1e-3) / 1e12 if admm_on else 0.0
    # compulsory HBM bytes per solve: x0 in, U + info out, plus the (z, w) state the kernels exchange
    hbm_bytes_per_solve = 8 * prob["nx"] + 8 * n + 16 + 4 * 8 * pl.n
    hbm_gbs = B * hbm_bytes_per_solve / (both_ms * 1e-3) / 1e9

    if admm_on:
        roofline = {"bound": "mfma", "kernel": "admm_shared_box_kernel<8,30>", "achieved": achieved_tf,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS,
                    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
                    # WRITE_SIZE in separate runs; FETCH doubled as the gfx950 guide prescribes), measured at 65,536
                    # instances and scaled with the batch; not re-measured by this script
                    "traffic": ADMM_TRAFFIC_BYTES_65536 * B / 65536.0,
                    "traffic_source": "profiles/r01_v5_pmc.csv (2*FETCH_SIZE + WRITE_SIZE of admm_shared_box_kernel)",
                    "algorithmic_bytes": B * (8 * prob["nx"] + 3 * 8 * 16 * ((n + 15) // 16)),
                    "flops_per_launch": admm_flops, "launch_ms": admm_ms}
    else:
        # default: no ADMM launch; the dominant kernel is the first active-set pass.  Its algorithmic bytes per instance:
        # x0 in; U, the (z, w) warm-start state and the info record out.  It is NOT an HBM-streaming kernel: a dependent
        # chain of fp64 VALU steps per instance -- the fraction is reported as the contract asks
        pa_bytes = B * (8 * prob["nx"] + 8 * n + 2 * 8 * n + 36)
        pa_gbs = pa_bytes / (pass_a_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "polish_oct_kernel<8,8,16,512> (first active-set pass)", "achieved": pa_gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pa_gbs / HBM_PEAK_GBS,
                    "traffic": POLISH_A_TRAFFIC_BYTES_65536 * B / 65536.0,
                    "traffic_source": "profiles/r01_v7_pmc.csv (2*FETCH_SIZE + WRITE_SIZE of the first polish_oct_kernel launch)",
                    "algorithmic_bytes": pa_bytes, "launch_ms": pass_a_ms,
                    "note": "latency/VALU bound by construction: ~8 dependent active-set steps per instance in fp64; "
                            "VALU busy fraction and instruction counts in DESIGN.md"}
    out = None
    if rank == 0:
        total = world * B * args.steps
        out = {
            "metric": "QP solves/sec (node) at batch 65,536, nx=12 nu=4 N=30",
            "value": total / dt, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: quadrotor hover MPC nx=12 nu=4 N=30 (n=120 box-constrained QP), "
                                   f"{B} random x0 per GPU, shared LTI model",
                       "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world} (independent shards)",
                       "algorithm": ("ADMM (shared (Hs+rho I)^-1, fp64 MFMA GEMM form) + " if admm_on else "") +
                                    "dual active-set kernels (8/16 lanes per instance, explicit S^-1 in registers, Hs^-1 in LDS)",
                       "admm_iters_mean": iters_mean, "admm_iters_max": iters_max,
                       "polish_steps_mean": psteps_mean, "polish_steps_max": psteps_max,
                       "solved_fraction": n_solved / B, "rho": pl.rho, "setup_ms": setup_ms,
                       "p50_batch_latency_ms": p50_ms, "p99_batch_latency_ms": p99_ms, "mean_batch_latency_ms": both_ms, "admm_kernel_ms": admm_ms, "polish_kernels_ms": polish_ms,
                       "first_pass_ms": pass_a_ms, "cold_order_ms": order_ms,
                       "collectives_in_step": "all_reduce(MIN) of the all-solved flag" if world > 1 else "none",
                       "allgather_ms": allgather_ms,
                       "hbm_GBps_compulsory": hbm_gbs, "hbm_frac": hbm_gbs / HBM_PEAK_GBS},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(prob, args.cpu_seconds)
            except Exception as e:      # the baseline is a report, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
