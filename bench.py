#!/usr/bin/env python3
"""bench.py -- benchmarks of the MPC QP hot path on MI355X.

Default = the headline metric (BASELINE.json): QP solves/sec at batch 65,536, nx=12 nu=4 N=30 (quadrotor hover MPC,
configs[2] = "config 3" in SURVEY.md's 1-based numbering), fp64, synthetic x0 (numpy default_rng(0)).  One "step" = one
pass of the hot path over the batch: mpcqp_solve (cold-start ordering kernels + dual active-set kernels) with x0 already
resident in HBM; the model set-up (assemble + factor) is outside the timed region because the model is shared and fixed
(LTI) -- it is reported as `setup_ms` (first call, with the one-off costs) and `setup_warm_ms` (again on the warm handle).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5] [--scaling strong|weak]

--gpus N > 1: if the ranks do not exist yet (WORLD_SIZE unset) this process starts them itself -- N fresh children under
torch.distributed.run, BEFORE it touches the GPU -- relays rank 0's JSON line and exits with their code.  Under the
driver's own torch.distributed.run the ranks already exist and it just runs as one of them.
Scaling (BASELINE.json configs[2]): "strong" (default) = the 65,536 x0 are SPLIT over the ranks, rank r
solves rows shard_range(65536, r, N); "weak" = 65,536 per GPU.  No data-path collective: instances are independent.  Every
step ends with the all-reduce of the convergence flag (MAX of the worst per-instance status, 0 = all solved); the all-gather of the solutions is timed separately
(config.allgather_ms).

Other workloads (same JSON schema, `config.workload` names them): --config 2 cart-pole 4,096 instances; --config 4
kinematic-bicycle LTV closed loop, 8,192 rollouts, one step = one closed-loop step of every rollout (default 1,000 steps);
--config 5 long-horizon N=100 with state box + polytopic rows, 256 instances.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 256 * 256 * 2.4          # 256 CUs x 256 B/clk (ds_read_b64 / b128, MI355X_MICROARCH.md, LDS) x 2.4 GHz = 157 TB/s
# counters of the dominant kernels at 65,536 instances from the committed PMC passes (tools/pmc_summary.py;
# HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE, KB -> bytes, the guide's gfx950 correction)
PMC = {
    "admm": ((2 * 49314.40 + 213303.90) * 1024.0, "profiles/r01_v5_pmc.csv (admm_shared_box_kernel)"),
    "pass_a": ((2 * 7575.72 + 195238.18) * 1024.0, "profiles/r02_pmc.csv (polish_oct_kernel<8, 8, 16, 512, true>, the first pass)"),
}
try:
    with open(os.path.join(ROOT, "profiles", "r03_pmc_headline.json")) as _f:      # written by tools/pmc_headline.py from the r03 passes
        PMC_R03 = json.load(_f)
except (OSError, ValueError):
    PMC_R03 = None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 50 (1000 for --config 4)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[c-1]; 3 = the headline workload")
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL batch (strong) / per-GPU batch (weak); default per config")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-sweep", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--keep-state", type=int, default=0, choices=[0, 1],
                    help="configs 2/3: 1 = every solve also leaves its (z, w) warm-start state in the handle (default 0: a cold "
                         "batch solve, SURVEY.md 8(d)'s 1,072 B/solve case; the other one is reported as with_state_ms)")
    ap.add_argument("--no-split", action="store_true", help="configs 2/3: solve block-diagonal models as a whole (split off)")
    ap.add_argument("--graph", action="store_true",
                    help="configs 2/3: every timed step replays the recorded solve (HIP graph, Planner.capture_solve) instead of "
                         "issuing its launches from the host; measured equal within 2 % at every batch size (graph_replay_ms, "
                         "shard_sweep_ms.graph_replay_ms are reported either way), so the default stays the host-issued chain")
    ap.add_argument("--rho", type=float, default=None)
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 1000 if args.config == 4 else 50
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


# ---- launcher: runs in a process that never touches the GPU -------------------------------------------------------------
def launch_command(gpus, argv, port, python=sys.executable):
    """The command line of the N-rank job (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [python, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_env(gpus, n_devices, environ):
    """Environment of the ranks.  Fewer devices than ranks (the one-GPU development box) selects the REHEARSAL mode: the
    ranks share devices and the collectives run on host tensors over gloo -- the same code path, no RCCL."""
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if n_devices < gpus:
        env.setdefault("MPC_BENCH_BACKEND", "gloo")
    return env


def needs_launch(args, environ):
    return args.gpus > 1 and "WORLD_SIZE" not in environ


def visible_gpus(environ, sysfs="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs the ranks will see, WITHOUT touching the HIP runtime (the launcher must stay a process that never initialised
    the GPU; torch.cuda.device_count() can fall back to hipGetDeviceCount): the visibility variables if set, else the KFD
    topology (nodes with SIMDs are GPUs); None if neither can be read."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    try:
        n = 0
        for node in os.listdir(sysfs):
            with open(os.path.join(sysfs, node, "properties")) as f:
                for line in f:
                    k, _, val = line.partition(" ")
                    if k == "simd_count" and int(val) > 0:
                        n += 1
        return n
    except (OSError, ValueError):
        return None


def launch(args, argv):
    ndev = visible_gpus(os.environ)
    if ndev is None:
        ndev = args.gpus            # cannot tell: assume one device per rank (RCCL); MPC_BENCH_BACKEND=gloo overrides
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = launch_command(args.gpus, argv, port)
    # children are fresh processes (never an exec of this one); stdout is inherited, so rank 0's JSON line is ours
    return subprocess.call(cmd, env=launch_env(args.gpus, ndev, os.environ))


# ---- CPU baselines (oracle/: checker and baseline only, never the product) ----------------------------------------------
def _host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        cores = int(os.environ["OMP_NUM_THREADS"])
    return max(1, min(cores, 16))


def cpu_baseline_box(prob, seconds, what):
    """oracle/qp_ipm.c (dense Mehrotra IPM on (P,q,G,h), cvxopt's default tolerances, one QP per thread) and
    oracle/qp_activeset.c (the GPU's own algorithm, exact) on a bounded sample of the SAME workload."""
    import numpy as np
    from oracle import cpu_baseline as cb
    from oracle import mpc_oracle as mo
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    cores = _host_cores()
    _, _, _, t1 = cb.solve_batch_box(H, F, f0, lb, ub, prob["x0"][:64], nthreads=1)
    per_core = 64 / t1
    nsamp = int(min(len(prob["x0"]), max(256, per_core * cores * seconds)))
    U, it, bad, t = cb.solve_batch_box(H, F, f0, lb, ub, prob["x0"][:nsamp], nthreads=cores)
    out = {"value": nsamp / t, "unit": "solves/s", "cores": cores, "kind": "port",
           "sample": f"first {nsamp} of the {len(prob['x0'])} x0 of {what} (same H,F,bounds); dense Mehrotra IPM, cvxopt "
                     f"default tolerances, mean {float(it.mean()):.1f} iterations, {bad} not optimal; "
                     f"single-core {per_core:.0f} solves/s",
           "single_core": per_core}
    # single-instance latency of the IPM port, one core: p50 over 101 solves of different x0
    lat = []
    for i in range(101):
        _, _, _, ti = cb.solve_batch_box(H, F, f0, lb, ub, prob["x0"][i:i + 1], nthreads=1)
        lat.append(ti)
    out["single_instance_p50_ms"] = sorted(lat)[50] * 1e3
    # the ADMM of admm_shared.hip as a NumPy restatement (batched GEMM form, OpenBLAS on all cores): a FIXED 200 iterations,
    # which reach eps 1e-3-1e-4 on this workload -- ADMM alone does not reach the 1e-6 bar in any sane budget
    try:
        ns3 = int(min(len(prob["x0"]), 4096)); K3 = 200
        fac = mo.factor(H, F, f0, lb, ub, 1.0)
        rho3 = 0.1 * float(np.sqrt(np.abs(fac["Hs"]).sum(1).max() / np.abs(fac["Hi"]).sum(1).max()))     # the rule of mpcqp_factor
        fac = mo.factor(H, F, f0, lb, ub, rho3)
        t3 = time.perf_counter(); _, _, rp, rd = mo.admm_box(fac, prob["x0"][:ns3], K3, return_residuals=True); t3 = time.perf_counter() - t3
        out["cpu_admm"] = {"value": ns3 / t3, "unit": "solves/s", "iterations": K3, "cores": cores,
                           "what": f"oracle/mpc_oracle.admm_box (NumPy, the GPU ADMM kernel's iteration) on {ns3} x0, {K3} "
                                   f"iterations each; residuals reached: primal {float(np.median(rp)):.1e}, dual {float(np.median(rd)):.1e} (median)"}
    except Exception as e:      # noqa: BLE001
        out["cpu_admm"] = {"value": None, "what": f"failed: {e!r}"}
    if hasattr(cb, "solve_batch_box_activeset"):
        try:
            ns2 = int(min(len(prob["x0"]), 8 * nsamp))
            _, st2, bad2, t2 = cb.solve_batch_box_activeset(H, F, f0, lb, ub, prob["x0"][:ns2], nthreads=cores)
            out["same_algorithm"] = {"value": ns2 / t2, "unit": "solves/s", "cores": cores,
                                     "what": "dual active-set method on the shared inverse (the GPU's algorithm, exact "
                                             f"minimiser) in C, OpenMP; first {ns2} x0, mean {float(st2.mean()):.1f} "
                                             f"steps, {bad2} failed"}
        except Exception as e:      # noqa: BLE001
            out["same_algorithm"] = {"value": None, "what": f"failed: {e!r}"}
    if hasattr(cb, "solve_batch_box_activeset_split"):
        try:
            ns2 = int(min(len(prob["x0"]), 8 * nsamp))
            _, st4, bad4, t4, nc4 = cb.solve_batch_box_activeset_split(H, F, f0, lb, ub, prob["x0"][:ns2], nthreads=cores)
            out["same_algorithm_split"] = {"value": ns2 / t4, "unit": "solves/s", "cores": cores, "components": int(nc4),
                                           "what": "the same C routine block by block where the Hessian splits (the GPU's "
                                                   f"component split, DESIGN.md 4b); first {ns2} x0, mean {float(st4.mean()):.1f} "
                                                   f"steps summed over {nc4} blocks, {bad4} failed"}
        except Exception as e:      # noqa: BLE001
            out["same_algorithm_split"] = {"value": None, "what": f"failed: {e!r}"}
    return out


# ---- collectives ---------------------------------------------------------------------------------------------------------
class Ranks:
    """One rank per GPU.  backend "nccl" = RCCL on device tensors; "gloo" (MPC_BENCH_BACKEND, rehearsal on a box with
    fewer GPUs than ranks) = host tensors."""

    def __init__(self):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        ndev = max(1, torch.cuda.device_count())
        self.local = local % ndev
        self.has_gpu = torch.cuda.device_count() > 0        # False only in the CPU tests of this class (gloo, host tensors)
        if self.has_gpu:
            torch.cuda.set_device(self.local)
        self.dev = torch.device(f"cuda:{self.local}" if self.has_gpu else "cpu")
        # MPC_BENCH_FORCE_DIST=1: initialise the process group even for one rank (exercises the RCCL calls on a one-GPU box)
        if self.world > 1 or (os.environ.get("MPC_BENCH_FORCE_DIST") and "RANK" in os.environ):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            self.backend = os.environ.get("MPC_BENCH_BACKEND", "nccl")
            if not dist.is_initialized():
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=self.dev)
                else:
                    dist.init_process_group(self.backend)
            self.dist = dist
        self.cdev = self.dev if self.backend in (None, "nccl") else torch.device("cpu")
        self._bar = torch.zeros(1, dtype=torch.int32, device=self.cdev)
        # the worst status of every step goes into one of two banks of 8 ints; a full bank travels in ONE all-reduce
        self._banks = [torch.zeros(8, dtype=torch.int32, device=self.cdev) for _ in range(2)]
        self._pending = [None, None]
        self._n = 0
        self._failed = None         # running MAX over the reduced banks (a tensor on cdev)
        self.all_ok = True
        if self.dist is not None:
            # first use of a collective on these buffers and of the two tiny torch kernels of _check loads code objects
            # (tens of milliseconds on ROCm): not inside the timed region
            for bank in (0, 1):
                self.dist.all_reduce(self._banks[bank], op=self.dist.ReduceOp.MAX)
                self._check(bank)
            self._failed = None
            if self.cdev.type == "cuda":
                torch.cuda.synchronize(self.dev)

    def barrier(self):
        # an all-reduce of one int synchronises the ranks on every backend (torch's gloo barrier() hung on the rehearsal
        # box in round 1; RCCL's own barrier is this all-reduce)
        if self.dist is not None:
            self.dist.all_reduce(self._bar)
            if self.cdev.type == "cuda":
                self.torch.cuda.synchronize(self.dev)

    def flag_allreduce(self, status_dev):
        """all-reduce(MAX) of every step's worst per-instance status (0 = every instance of every rank solved: the
        convergence flag).  Bucketed: the flags of 8 consecutive steps are reduced by one collective (measured with one
        rank on RCCL, tools/collective_cost.py: an asynchronous all-reduce per step costs the compute stream 23 us whatever
        its size -- a fifth of an 8,192-instance step -- the reduction kernel that writes the flag 5-14 us).  Two banks: a
        bank is only written again once the collective that read it has finished (16 steps later it long has).  drain()
        ends the timed region, so every collective is inside it."""
        if self.dist is None:
            return
        bank, j = (self._n >> 3) & 1, self._n & 7
        if j == 0 and self._pending[bank] is not None:
            self._pending[bank].wait()
            self._check(bank)
            self._pending[bank] = None
        f = self._banks[bank][j:j + 1]
        if self.cdev.type == "cuda":
            # RCCL: the reduction kernel writes the flag buffer itself, on the compute stream, which the collective's stream waits for
            self.torch.amax(status_dev, dim=0, keepdim=True, out=f)
        else:
            # gloo: a BLOCKING copy to the host tensor -- it must have landed before the collective reads it
            f.copy_(self.torch.amax(status_dev, dim=0, keepdim=True))
        self._n += 1
        if j == 7:
            self._pending[bank] = self.dist.all_reduce(self._banks[bank], op=self.dist.ReduceOp.MAX, async_op=True)

    def _check(self, bank):
        # a reduced bank: any non-zero entry is a step in which some instance of some rank was not solved
        bad = self._banks[bank].amax()
        self._failed = bad if self._failed is None else self.torch.maximum(self._failed, bad)

    def drain(self):
        if self.dist is None or not self._n:
            return
        if self._n & 7:         # the partly filled bank (its other entries are zero or already-reduced values)
            bank = (self._n >> 3) & 1
            self._pending[bank] = self.dist.all_reduce(self._banks[bank], op=self.dist.ReduceOp.MAX, async_op=True)
        for bank, w in enumerate(self._pending):
            if w is not None:
                w.wait()
                self._check(bank)
                self._pending[bank] = None
        if self.cdev.type == "cuda":
            self.torch.cuda.synchronize(self.dev)
        self.all_ok = self.all_ok and (self._failed is None or int(self._failed.item()) == 0)

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()


def device_timer(torch, dev):
    """ms per call from events on the stream the kernels are launched on (the Planner uses torch's current stream)."""
    def timed(fn, reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps
    return timed


# ---- workloads -------------------------------------------------------------------------------------------------------------
class BoxWorkload:
    """configs[1] (cart-pole) and configs[2] (quadrotor, the headline): shared LTI model, box-constrained inputs."""

    def __init__(self, args, ranks):
        import torch
        from mpc_amd.configs import make_config
        from mpc_amd.distributed import shard_range
        from mpc_amd.planner import Planner
        self.args, self.ranks, self.torch = args, ranks, torch
        cfg = args.config
        default_batch = {2: 4096, 3: 65536}[cfg]
        G = args.batch or default_batch
        if args.scaling == "strong":
            self.global_batch = G
            self.prob = make_config(cfg, batch=G, seed=0)           # every rank draws the SAME global set ...
            lo, hi = shard_range(G, ranks.rank, ranks.world)        # ... and solves its contiguous row block
        else:
            self.global_batch = G * ranks.world
            self.prob = make_config(cfg, batch=G, seed=ranks.rank)  # every rank draws its own shard
            lo, hi = 0, G
        prob = self.prob
        self.B = hi - lo
        dev = ranks.dev
        self.pl = pl = Planner(prob["nx"], prob["nu"], prob["N"], device=dev)
        pl.split = not args.no_split
        pl.opts.keep_state = args.keep_state
        pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.set_constraints(prob["u_min"], prob["u_max"])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pl.setup(rho=args.rho)
        torch.cuda.synchronize(dev)
        self.setup_ms = (time.perf_counter() - t0) * 1e3
        # the same set-up again on the warm handle (code objects loaded, workspaces allocated): what a caller who changes the
        # weights between batches pays per change; the first call above also carries the one-off costs
        ts = []
        for _ in range(3):
            pl.set_cost(prob["Q"], prob["R"], prob["P"])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            pl.setup(rho=args.rho)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e3)
        self.setup_warm_ms = min(ts)
        self.x0_all = torch.as_tensor(prob["x0"], device=dev)
        self.x0 = self.x0_all[lo:hi].contiguous()
        self.opts = {} if args.max_iter is None else {"max_iter": args.max_iter}
        pl.reuse_outputs = True
        self.U = None
        # the solve's launch chain recorded once (Planner.capture_solve) and replayed: one graph launch per batch instead of
        # five to seven kernel launches.  Timed beside the host-issued chain on one GPU; the timed step uses it under --graph
        self.graph, self.graph_error = None, None
        if args.graph or ranks.world == 1:
            try:
                self.graph = pl.capture_solve(self.x0, **self.opts)
            except Exception as e:      # noqa: BLE001 -- a failed capture must not lose the line: the eager path is the same solve
                self.graph_error = repr(e)
        self.units_per_step = self.global_batch
        self.name = {2: "configs[1]: cart-pole linearised MPC nx=4 nu=1 N=50 (n=50 box-constrained QP)",
                     3: "configs[2]: quadrotor hover MPC nx=12 nu=4 N=30 (n=120 box-constrained QP)"}[cfg]

    def step(self):
        # reuse_outputs: the same output buffers every step (an eager call otherwise spends more host time on small
        # allocations than the device needs for a 1/8 shard); the convergence flag is one reduction kernel
        self.U = self.graph.replay() if (self.graph is not None and self.args.graph) else self.pl.solve(self.x0, **self.opts)
        if self.ranks.dist is not None:
            self.ranks.flag_allreduce(self.pl.info.status)

    def report(self, dt):
        import numpy as np
        args, ranks, torch, pl, prob = self.args, self.ranks, self.torch, self.pl, self.prob
        dev, B, x0, opts = ranks.dev, self.B, self.x0, self.opts
        timed = device_timer(torch, dev)
        info = pl.info
        n_solved = int((info.status == 0).sum().item())
        iters_mean = float(info.iters.float().mean().item()); iters_max = int(info.iters.max().item())
        psteps_mean = float(info.polish_steps.float().mean().item()); psteps_max = int(info.polish_steps.max().item())
        reps = max(5, min(args.steps, 20))
        admm_on = iters_max > 0
        admm_ms = timed(lambda: pl.solve(x0, stages=1, **opts), reps) if admm_on else 0.0
        both_ms = timed(lambda: pl.solve(x0, **opts), reps)
        graph_ms = timed(self.graph.replay, reps) if self.graph is not None else None
        # the first (and longest) active-set launch alone: stages bit 2 stops after it, bit 3 after the cold-start
        # ordering kernels that run before it (diagnostic modes: queued instances stay open)
        # (the ordering kernel only runs for cold batches of more than 16,384 instances, polish_quad.hip; below that "stages=15"
        # times an empty solve, i.e. the host's launch cost, which is not something to subtract from a kernel time)
        split = pl.split_info()
        ncomp = split["components"]
        ordered = (B > 16384 or ncomp > 0) and not admm_on
        order_ms = max(timed(lambda: pl.solve(x0, stages=15, **opts), reps) - admm_ms, 0.0) if ordered else 0.0
        pass_a_ms = max(timed(lambda: pl.solve(x0, stages=7, **opts), reps) - admm_ms - order_ms, 0.0)
        nlat = max(20, min(args.steps, 100))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlat)]
        for e0, e1 in evs:
            e0.record(); pl.solve(x0, **opts); e1.record()
        torch.cuda.synchronize(dev)
        lat = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        p50_ms, p99_ms = lat[len(lat) // 2], lat[min(len(lat) - 1, int(0.99 * len(lat)))]

        allgather_ms = None
        if ranks.dist is not None and ranks.cdev == dev and args.scaling == "strong":
            from mpc_amd.distributed import allgather_solutions
            allgather_ms = timed(lambda: allgather_solutions(self.U, self.global_batch), 5)

        # shard sizes of the 2/4/8-GPU split of the same batch, on THIS GPU: what strong scaling will be made of
        sweep = None
        if ranks.world == 1 and not args.no_shard_sweep and args.config == 3:
            sweep = {}
            eager = {}
            for div in (1, 2, 4, 8):
                b = self.global_batch // div
                xs = self.x0_all[:b].contiguous()
                sweep[str(b)] = timed(lambda: pl.solve(xs, **opts), reps)
                if self.graph is not None:
                    g = pl.capture_solve(xs, **opts)
                    eager[str(b)] = timed(g.replay, reps)
                    g.close()
            t1, t8 = sweep[str(self.global_batch)], sweep[str(self.global_batch // 8)]
            sweep["implied_strong_efficiency_8"] = t1 / (8.0 * t8)
            sweep["graph_replay_ms"] = eager or None
            pl.solve(x0, **opts)

        # the serving pattern: two handles on two streams, solves alternating between them, so that one solve's short
        # launches (ordering, the second pass's 100 workgroups, the empty passes) run beside the other's first pass.
        # A diagnostic, never `value`: a step of the contract is one solve on one stream
        overlap_ms = None
        if ranks.world == 1 and not args.no_shard_sweep and args.config == 3:
            from mpc_amd.planner import Planner
            pl2 = Planner(prob["nx"], prob["nu"], prob["N"], device=dev)
            pl2.split = pl.split; pl2.opts.keep_state = args.keep_state
            pl2.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
            pl2.set_constraints(prob["u_min"], prob["u_max"]).setup(rho=args.rho)
            pl2.reuse_outputs = True
            pls, streams = (pl, pl2), (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
            def alternate(k):
                for i in range(k):
                    with torch.cuda.stream(streams[i & 1]):
                        pls[i & 1].solve(x0, **opts)
            torch.cuda.synchronize(dev)
            alternate(4); torch.cuda.synchronize(dev)
            k = 2 * max(10, reps)
            t0 = time.perf_counter(); alternate(k); torch.cuda.synchronize(dev)
            overlap_ms = (time.perf_counter() - t0) * 1e3 / k
            ok2 = int((pl2.info.status == 0).sum().item())
            if ok2 != B:
                raise SystemExit(f"two-stream run: {B - ok2} instances of the second handle not solved")
            del pl2

        n = pl.n
        # the other state mode, beside `value`'s
        pl.opts.keep_state = 1 - args.keep_state
        other_ms = timed(lambda: pl.solve(x0, **opts), reps)
        pl.opts.keep_state = args.keep_state
        pl.solve(x0, **opts)
        # ---- three fractions for the timed region; the binding roofline is the largest ------------------
        # work actually executed by the active-set method, from the solution itself: a component (the whole instance when the
        # model does not split) of r rows that ends with a active rows took ~a add steps at working-set sizes k = 0..a-1, each
        #   x-update 2 r (k+1) + direction 2 k^2 + rank-one update of S^-1 2 (k+1)^2 + scan 3 r   flops,
        #   8 r (k+1) bytes of Hi rows read from LDS for the x-update (the other LDS reads are O(k)).
        lbv = torch.as_tensor(np.tile(prob["u_min"], prob["N"]), device=dev); ubv = torch.as_tensor(np.tile(prob["u_max"], prob["N"]), device=dev)
        Uf = self.U.reshape(B, -1)
        act = ((Uf >= ubv - 1e-9 * ubv.abs().clamp_min(1.0)) | (Uf <= lbv + 1e-9 * lbv.abs().clamp_min(1.0))).double()
        if ncomp > 0:
            rm = torch.as_tensor(split["row_map"], device=dev)
            a_c = torch.stack([act[:, rm == c].sum(1) for c in range(ncomp)], 1)      # active rows per (instance, component)
            r_c = float(split["rows"])
        else:
            a_c = act.sum(1, keepdim=True); r_c = float(n)
        as_flops = float((r_c * a_c * (a_c + 1) + (4.0 / 3.0) * a_c ** 3 + 2.0 * a_c ** 2 + 3.0 * r_c * (a_c + 1)).sum().item())
        as_flops += B * (2.0 * n * (prob["nx"] + 1))                                    # x_unc = Ku1 [x0; 1]
        lds_bytes = float((4.0 * r_c * a_c * (a_c + 1)).sum().item())
        t_s = both_ms * 1e-3
        compulsory = B * (8 * prob["nx"] + 8 * n + 16)                                   # x0 in; U, (iters, status, objective) out
        state_bytes = B * 2 * 8 * 16 * ((n + 15) // 16) if args.keep_state else 0       # (z, w) written when the state is kept
        fr_hbm = (compulsory + state_bytes) / t_s / 1e9 / HBM_PEAK_GBS
        fr_fp64 = (as_flops if not admm_on else as_flops + B * iters_mean * (2 * n * n + 10 * (n + n))) / t_s / 1e12 / FP64_PEAK_TFLOPS
        fr_lds = lds_bytes / t_s / 1e9 / LDS_PEAK_GBS
        fractions = {"hbm": fr_hbm, "fp64": fr_fp64, "lds": fr_lds,
                     "binding": max((("hbm", fr_hbm), ("fp64", fr_fp64), ("lds", fr_lds)), key=lambda kv: kv[1])[0],
                     "bytes_per_solve_compulsory": compulsory / B, "bytes_per_solve_state": state_bytes / B,
                     "flops_per_solve_executed": as_flops / B, "lds_bytes_per_solve": lds_bytes / B,
                     "what": "SURVEY.md 8(d): whole timed region (one mpcqp_solve of the batch); hbm = compulsory bytes (x0 in, U + "
                             "info out, + the (z, w) state when kept) / t / 8 TB/s; fp64 = flops the active-set method executed "
                             "(from the final active sets, formula in bench.py) / t / 78.6 TF; lds = Hi bytes its x-updates read / t "
                             "/ 157 TB/s.  All three are small: the solve is a dependent chain of short fp64 steps per (component of "
                             "an) instance -- VALU-issue and latency bound (valu_active / waves_per_simd below), not a streaming or a "
                             "GEMM-shaped kernel"}
        if admm_on:
            flops = B * iters_mean * (2 * n * n + 10 * (n + n))         # 2 n^2 + 10 (n + m) per iteration
            tf = flops / (admm_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "admm_shared_box_kernel", "achieved": tf, "peak": FP64_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "traffic": PMC["admm"][0] * B / 65536.0,
                        "traffic_source": PMC["admm"][1] + ", scaled with the batch",
                        "algorithmic_bytes": B * (8 * prob["nx"] + 3 * 8 * 16 * ((n + 15) // 16)),
                        "flops_per_launch": flops, "launch_ms": admm_ms, "fractions": fractions}
        else:
            # dominant kernel: the first active-set pass.  Its algorithmic bytes: x0 in, U (and the state when kept) out, of the
            # instances (components) it finishes -- the same per-solve figure, since every instance passes through it or through
            # the ordering kernel in front of it.  NOT a streaming kernel: the fraction is reported because the contract asks for one
            pa_bytes = compulsory + state_bytes
            gbs = pa_bytes / (pass_a_ms * 1e-3) / 1e9
            key = "split" if ncomp > 0 else "unsplit"
            pm = (PMC_R03 or {}).get(key) if args.config == 3 else None
            roofline = {"bound": "hbm", "kernel": ("polish_oct_kernel<8 lanes, 8 slots, split> (first active-set pass over the components)" if ncomp > 0
                                                   else "polish_oct_kernel (first active-set pass)"), "achieved": gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": (pm["hbm_bytes_per_launch"] * B / 65536.0 if pm else (PMC["pass_a"][0] * B / 65536.0 if args.config == 3 and ncomp == 0 and args.keep_state else None)),
                        "traffic_source": (pm["source"] + ", scaled with the batch" if pm else (PMC["pass_a"][1] + ", scaled with the batch" if args.config == 3 and ncomp == 0 and args.keep_state else None)),
                        "algorithmic_bytes": pa_bytes, "launch_ms": pass_a_ms,
                        "valu_active_per_simd": pm.get("valu_active_per_simd") if pm else None,
                        "waves_per_simd": pm.get("waves_per_simd") if pm else None,
                        "fractions": fractions,
                        "note": "latency / VALU-issue bound by construction: dependent active-set steps in fp64 (DESIGN.md section 4b)" +
                                ("" if B >= 16384 else "; at this batch size the launch is shorter than the host's cost of "
                                 "issuing it (~20 us per launch pair): launch_ms is host-bound")}
        config = {"workload": f"{self.name}, {self.global_batch} random x0 "
                              f"{'split over' if args.scaling == 'strong' else 'per GPU on'} {ranks.world} GPU(s), "
                              "shared LTI model",
                  "batch_per_gpu": B, "global_batch": self.global_batch,
                  "parallelism": f"dp{ranks.world} (independent shards, no data-path collective)",
                  "algorithm": ("ADMM (shared (Hs+rho I)^-1, fp64 MFMA GEMM form) + " if admm_on else "") +
                               (f"component split ({ncomp} independent blocks of {split['rows']} rows found in the Hessian): x_unc on MFMA "
                                "tiles finishes the untouched components, dual active-set kernels take the rest (8 lanes per "
                                "component, explicit S^-1 in registers, the blocks of Hs^-1 in LDS)" if ncomp > 0 and not admm_on else
                                "dual active-set kernels (8/32 lanes per instance, explicit S^-1 in registers, Hs^-1 in LDS)"),
                  "split_components": ncomp, "keep_state": args.keep_state,
                  ("with_state_ms" if not args.keep_state else "without_state_ms"): other_ms,
                  "admm_iters_mean": iters_mean, "admm_iters_max": iters_max,
                  "polish_steps_mean": psteps_mean, "polish_steps_max": psteps_max,
                  "solved_fraction": n_solved / B, "rho": pl.rho, "setup_ms": self.setup_ms, "setup_warm_ms": self.setup_warm_ms,
                  "p50_batch_latency_ms": p50_ms, "p99_batch_latency_ms": p99_ms, "mean_batch_latency_ms": both_ms,
                  "admm_kernel_ms": admm_ms, "polish_kernels_ms": max(both_ms - admm_ms, 0.0),
                  "first_pass_ms": pass_a_ms, "cold_order_ms": order_ms,
                  "step_issued_as": "graph replay (Planner.capture_solve)" if (self.graph is not None and args.graph) else "host launches",
                  "graph_replay_ms": graph_ms, "graph_capture_error": self.graph_error,
                  "collectives_in_step": "all_reduce(MAX) of the worst status of every step (0 = all solved), 8 steps per collective" if ranks.world > 1 else "none",
                  "collective_backend": ranks.backend, "allgather_ms": allgather_ms, "shard_sweep_ms": sweep,
                  "two_handles_two_streams_ms_per_solve": overlap_ms,
                  "hbm_GBps_compulsory": compulsory / (both_ms * 1e-3) / 1e9}
        return config, roofline

    def cpu_baseline(self):
        return cpu_baseline_box(self.prob, self.args.cpu_seconds, self.name.split(":")[0])


class LtvWorkload:
    """configs[3]: kinematic-bicycle LTV closed loop with warm-start shifting; one model per rollout, re-linearised,
    assembled, factored and solved every step.  One step = one closed-loop step of every rollout = `rollouts` QPs."""

    def __init__(self, args, ranks):
        import torch
        from mpc_amd.configs import make_config
        from mpc_amd.distributed import shard_range
        from mpc_amd.ltv import Bicycle, ClosedLoopLTV
        self.args, self.ranks, self.torch = args, ranks, torch
        G = args.batch or 8192
        if args.scaling == "strong":
            self.global_batch = G
            cfg = make_config(4, batch=G, seed=0)
            lo, hi = shard_range(G, ranks.rank, ranks.world)
        else:
            self.global_batch = G * ranks.world
            cfg = make_config(4, batch=G, seed=ranks.rank)
            lo, hi = 0, G
        self.cfg = cfg
        self.B = hi - lo
        dev = ranks.dev
        self.N, self.dt = cfg["N"], cfg["dt"]
        self.loop = ClosedLoopLTV(Bicycle(cfg["L"], self.dt), self.N, cfg["Q"], cfg["R"], cfg["P"], cfg["u_min"],
                                  cfg["u_max"], device=dev)
        self.x = torch.as_tensor(cfg["x0"][lo:hi], device=dev).contiguous()
        self.ks = torch.arange(1, self.N + 1, device=dev, dtype=torch.float64)
        self.ahead = (cfg["v_ref"] * self.dt * self.ks).reshape(1, self.N)      # reference positions ahead of the rollout, (1, N)
        self.x_ref = torch.zeros((self.B, self.N, 4), dtype=torch.float64, device=dev)
        self.x_ref[:, :, 3] = cfg["v_ref"]
        self.units_per_step = self.global_batch
        self.setup_ms = 0.0
        self.nsteps = 0

    def step(self):
        cfg = self.cfg
        self.torch.add(self.x[:, :1], self.ahead, out=self.x_ref[:, :, 0])             # one launch: x + v_ref dt k into the strided column
        _, self.x = self.loop.step(self.x, self.x_ref, warm=True, advance=True)    # the plant's step rides in the apply launch
        self.nsteps += 1
        if self.ranks.dist is not None:
            self.ranks.flag_allreduce(self.loop.pl.info.status)

    def report(self, dt):
        import ctypes as C
        args, ranks, torch = self.args, self.ranks, self.torch
        dev = ranks.dev
        pl = self.loop.pl
        timed = device_timer(torch, dev)
        info = pl.info
        solved = float((info.status == 0).float().mean().item())
        st_mean = float(info.polish_steps.float().mean().item())
        # per-kernel device time of one step's library calls, on the data of the last step
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        lib, h = pl._lib, pl._h
        from mpc_amd import _lib
        fo = _lib.FactorOpts(0.0, 1, 1)
        asm_ms = timed(lambda: lib.mpcqp_assemble(h, stream), 20)
        fac_ms = timed(lambda: (lib.mpcqp_assemble(h, stream), lib.mpcqp_factor(h, C.byref(fo), stream)), 20) - asm_ms
        nx, nu, N = 4, 2, self.N
        n, p = N * nu, N * nx
        # the dominant kernel of a step is the per-model factorisation (factor_wave_kernel: Jacobi scaling, blocked Cholesky,
        # explicit inverse, affine maps).  Algorithmic flops per model: Cholesky n^3/3 + triangular inverse n^3/3 + Hi = T'T
        # n^3/3 ... counted as multiply-add pairs x 2 = 2 n^3 in all, + the maps 2 n^2 (nx + 1)
        flops = 2 * n ** 3 + 2 * n * n * (nx + 1)
        tf = self.B * flops / (fac_ms * 1e-3) / 1e12
        # SURVEY.md 8(d) assembly flops per distinct system, prediction-matrix form (rollout + Qbar Gamma + H as SYRK + F); the
        # structured assembly that runs (assemble_wave_kernel) needs about a sixth of them and writes neither Gamma nor Qbar Gamma
        asm_flops = (2 * nx ** 3 * N + 2 * nx * nx * nu * (N - 1)) + 2 * nx * nx * nu * N * (N + 1) // 2 + p * n * (n + 1) \
            + 2 * n * p * nx
        roofline = {"bound": "mfma", "kernel": "factor_wave_kernel (one wave per rollout's model)", "achieved": tf,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "traffic": None,
                    "flops_per_launch": self.B * flops, "launch_ms": fac_ms,
                    "assemble_TFLOPs_prediction_matrix_form": self.B * asm_flops / (asm_ms * 1e-3) / 1e12,
                    "note": "n = 40 (padded to 48): 16 x 16 MFMA tiles inside dependent chains, 8 waves per SIMD in turn -- bound "
                            "by instruction issue, far from the MFMA peak by construction (DESIGN.md section 4)"}
        config = {"workload": f"configs[3]: kinematic-bicycle LTV MPC closed loop nx=4 nu=2 N={N} (n={n}), "
                              f"{self.global_batch} parallel rollouts over {ranks.world} GPU(s), warm-start shifting, every "
                              "rollout re-linearised, assembled, factored and solved each step",
                  "batch_per_gpu": self.B, "global_batch": self.global_batch, "closed_loop_steps": args.steps,
                  "parallelism": f"dp{ranks.world} (independent rollouts)",
                  "ms_per_closed_loop_step": dt / args.steps * 1e3, "assemble_ms": asm_ms, "factor_ms": fac_ms,
                  "active_set_steps_mean_last": st_mean, "solved_fraction_last": solved,
                  "final_abs_y_max": float(self.x[:, 1].abs().max().item()),
                  "collectives_in_step": "all_reduce(MAX) of the worst status of every step (0 = all solved), 8 steps per collective" if ranks.world > 1 else "none",
                  "collective_backend": ranks.backend}
        return config, roofline

    def cpu_baseline(self):
        """The same closed-loop QP on the host: NumPy condensing (oracle/mpc_oracle.py) + the C IPM per rollout, on a
        sample of the rollouts of the last step (same linearisation data, read back from the device)."""
        import numpy as np
        from oracle import cpu_baseline as cb
        from oracle import mpc_oracle as mo
        last = self.loop.last
        cores = _host_cores()
        ns = min(self.B, 64 * cores)
        A = last["A"][:ns].cpu().numpy(); Bm = last["B"][:ns].cpu().numpy()
        buf = self.loop._buf
        xr = buf["xr"][:ns].cpu().numpy(); ur = buf["ur"][:ns].cpu().numpy()
        lb = buf["lb"][:ns].cpu().numpy().reshape(ns, -1); ub = buf["ub"][:ns].cpu().numpy().reshape(ns, -1)
        cfg = self.cfg
        t0 = time.perf_counter()
        bad = 0
        for i in range(ns):
            prob = dict(N=self.N, nx=4, nu=2, A=A[i], B=Bm[i], Q=cfg["Q"], R=cfg["R"], P=cfg["P"])
            H, F, f0 = mo.condense(prob, x_ref=xr[i], u_ref=ur[i])
            _, _, b, _ = cb.solve_batch_box(H, F, f0, lb[i], ub[i], np.zeros((1, 4)), nthreads=1)
            bad += b
        t = time.perf_counter() - t0
        return {"value": ns / t, "unit": "solves/s", "cores": 1, "kind": "port",
                "sample": f"{ns} rollouts of the last closed-loop step: NumPy condensing + C Mehrotra IPM per rollout, "
                          f"single thread, {bad} not optimal"}


class RowsWorkload:
    """configs[4]: long horizon N=100, nx=20, nu=6 (n=600) with input box + state box + polytopic rows (13,200 one-sided
    rows), shared model, constraint-space active-set kernel."""

    def __init__(self, args, ranks):
        import torch
        from mpc_amd.configs import make_config
        from mpc_amd.distributed import shard_range
        from mpc_amd.planner import Planner
        self.args, self.ranks, self.torch = args, ranks, torch
        G = args.batch or 256
        if args.scaling == "strong":
            self.global_batch = G
            prob = make_config(5, batch=G, seed=0)
            lo, hi = shard_range(G, ranks.rank, ranks.world)
        else:
            self.global_batch = G * ranks.world
            prob = make_config(5, batch=G, seed=ranks.rank)
            lo, hi = 0, G
        self.prob = prob
        self.B = hi - lo
        dev = ranks.dev
        self.pl = pl = Planner(prob["nx"], prob["nu"], prob["N"], device=dev)
        pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"], C=prob["C"],
                           c_min=prob["c_min"], c_max=prob["c_max"])
        self.x0 = torch.as_tensor(prob["x0"][lo:hi], device=dev).contiguous()
        pl.reuse_outputs = True
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pl.setup()                      # assemble + the rows and their dual Hessian (mpcqp_factor, general mode)
        pl.solve(self.x0[:1])
        torch.cuda.synchronize(dev)
        self.setup_ms = (time.perf_counter() - t0) * 1e3
        self.units_per_step = self.global_batch

    def step(self):
        self.U = self.pl.solve(self.x0)
        if self.ranks.dist is not None:
            self.ranks.flag_allreduce(self.pl.info.status)

    def report(self, dt):
        args, ranks, torch, pl, prob = self.args, self.ranks, self.torch, self.pl, self.prob
        timed = device_timer(torch, ranks.dev)
        info = pl.info
        n, m = pl.n, int(pl.general_info()["m"])
        steps = info.polish_steps.double()
        k_mean, k_max = float(steps.mean().item()), int(steps.max().item())
        solve_ms = timed(lambda: pl.solve(self.x0), 10)
        # algorithmic bytes of the active-set kernel per instance: step s reads s+1 rows of the
        # dual Hessian and reads + writes the m row residuals (the scan for the next row rides on that pass; the first scan
        # reads them once); the tail reads the k active rows of Cs (twice) and of CH.  No n x n matrix is read per instance
        # since the tail certifies optimality from the set-up's |I - Ps Hi|
        per_inst = float((8.0 * m * (steps * (steps + 1) / 2 + 3 * steps + 1) + 8.0 * 3 * n * steps).mean().item()) + 8.0 * 2 * m
        gbs = self.B * per_inst / (solve_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "rows_qp_kernel + its GEMM prologue (whole mpcqp_rows_solve_mpc call)",
                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes": self.B * per_inst, "launch_ms": solve_ms}
        config = {"workload": f"configs[4]: long-horizon MPC nx={prob['nx']} nu={prob['nu']} N={prob['N']} (n={n}, {m} one-sided "
                              f"rows: input box + state box + polytope), {self.global_batch} x0 over {ranks.world} GPU(s), "
                              "shared model",
                  "batch_per_gpu": self.B, "global_batch": self.global_batch, "path": pl.general_info()["kernel"],
                  "parallelism": f"dp{ranks.world} (independent shards)", "setup_ms": self.setup_ms,
                  "active_set_steps_mean": k_mean, "active_set_steps_max": k_max,
                  "solved_fraction": float((info.status == 0).float().mean().item()), "batch_latency_ms": solve_ms,
                  "collectives_in_step": "all_reduce(MAX) of the worst status of every step (0 = all solved), 8 steps per collective" if ranks.world > 1 else "none",
                  "collective_backend": ranks.backend}
        return config, roofline

    def cpu_baseline(self):
        import numpy as np
        from oracle import cpu_baseline as cb
        pl = self.pl
        H, F, f0, G, E, h0 = (pl.get(k).cpu().numpy() for k in ("H", "F", "f0", "G", "E", "h0"))
        x0 = self.prob["x0"]
        from concurrent.futures import ThreadPoolExecutor
        cores = _host_cores()
        ns = min(cores, len(x0))            # one instance per thread (ctypes releases the GIL), ~30 s each at full size
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(lambda i: cb.solve_dense(H, F @ x0[i] + f0, G, h0 - E @ x0[i]), range(ns)))
        t = time.perf_counter() - t0
        bad = sum(int(r[2] != 0) for r in res)
        return {"value": ns / t, "unit": "solves/s", "cores": cores, "kind": "port",
                "sample": f"{ns} of the {len(x0)} instances (n={H.shape[0]}, {G.shape[0]} rows), one per thread: C Mehrotra "
                          f"IPM, cvxopt default tolerances, mean {np.mean([r[3] for r in res]):.1f} iterations, "
                          f"{bad} not optimal"}


WORKLOADS = {2: BoxWorkload, 3: BoxWorkload, 4: LtvWorkload, 5: RowsWorkload}
METRIC = {2: "QP solves/sec at batch 4,096, nx=4 nu=1 N=50 (cart-pole)",
          3: "QP solves/sec (node) at batch 65,536, nx=12 nu=4 N=30",
          4: "QP solves/sec, kinematic-bicycle LTV closed loop, 8,192 rollouts",
          5: "QP solves/sec, long-horizon N=100 nx=20 nu=6 with state box + polytopic rows"}


def run_rank(args):
    import torch
    ranks = Ranks()
    dev = ranks.dev
    verbose = bool(os.environ.get("MPC_BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print(f"[rank {ranks.rank}] {msg}", file=sys.stderr, flush=True)

    wl = WORKLOADS[args.config](args, ranks)
    note("setup done")
    for _ in range(args.warmup):
        wl.step()
    ranks.drain()
    note("warmup done")
    ranks.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    ranks.drain()
    ranks.barrier()
    torch.cuda.synchronize(dev)
    dt = ranks.max_over_ranks(time.perf_counter() - t0)
    note("timed region done")
    config, roofline = wl.report(dt)
    note("per-kernel timing done")
    if ranks.rank == 0:
        out = {"metric": METRIC[args.config], "value": wl.units_per_step * args.steps / dt, "unit": "solves/s",
               "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": config, "roofline": roofline}
        if ranks.dist is not None:
            out["config"]["all_ranks_all_solved_every_step"] = bool(ranks.all_ok)      # what the flag collectives returned
        if ranks.world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = wl.cpu_baseline()
            except Exception as e:      # noqa: BLE001 -- the baseline is a report, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    ranks.finish()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if needs_launch(args, os.environ):
        sys.exit(launch(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
