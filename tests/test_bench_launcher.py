"""bench.py --gpus N: the parent starts N fresh ranks before touching the GPU (CPU test of that logic) and the
partition it measures is BASELINE.json's split of the batch over the ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_defaults_and_launch_decision():
    b = _bench()
    a = b.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.config, a.scaling) == (1, 50, 5, 3, "strong")
    assert b.parse_args(["--config", "4"]).steps == 1000          # BASELINE.json configs[3]: 1,000 closed-loop steps
    a8 = b.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert b.needs_launch(a8, {}) is True                           # `python bench.py --gpus 8` typed by hand
    assert b.needs_launch(a8, {"WORLD_SIZE": "8", "RANK": "0"}) is False   # already a rank of torch.distributed.run
    assert b.needs_launch(a, {}) is False
    with pytest.raises(SystemExit):
        b.parse_args(["--gpus", "0"])


def test_launch_command_and_environment(tmp_path):
    b = _bench()
    argv = ["--gpus", "4", "--steps", "7"]
    cmd = b.launch_command(4, argv, 29511, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    env = b.launch_env(4, 8, {"PATH": "x"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "MPC_BENCH_BACKEND" not in env      # enough GPUs: RCCL
    assert b.launch_env(2, 1, {})["MPC_BENCH_BACKEND"] == "gloo"                           # one-GPU box: rehearsal
    assert b.launch_env(2, 1, {"MPC_BENCH_BACKEND": "nccl"})["MPC_BENCH_BACKEND"] == "nccl"
    # the launcher counts GPUs without the HIP runtime: visibility variables first, then the KFD topology
    assert b.visible_gpus({"HIP_VISIBLE_DEVICES": "0,3"}) == 2 and b.visible_gpus({"ROCR_VISIBLE_DEVICES": ""}) == 0
    top = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024)):                  # node 0 is the CPU
        (top / str(i)).mkdir(parents=True)
        (top / str(i) / "properties").write_text(f"cpu_cores_count {8 if i == 0 else 0}\nsimd_count {simd}\n")
    assert b.visible_gpus({}, sysfs=str(top)) == 2
    assert b.visible_gpus({}, sysfs=str(tmp_path / "missing")) is None
    import ast
    src = open(b.__file__).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "launch")
    assert "torch" not in ast.unparse(fn)                       # no torch (hence no HIP) in the launcher process


def test_launcher_starts_fresh_ranks_and_relays_rank0(tmp_path):
    """The same launch path with a stand-in rank script: N processes under torch.distributed.run on 127.0.0.1, a gloo
    all-reduce between them, rank 0's JSON line on the parent's stdout, the children's exit code returned."""
    b = _bench()
    script = tmp_path / "rank.py"
    script.write_text(
        "import json, os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([dist.get_rank() + 1]); dist.all_reduce(t)\n"
        "if dist.get_rank() == 0: print(json.dumps({'n_gpus': dist.get_world_size(), 'sum': int(t), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.destroy_process_group()\n"
        "sys.exit(3 if '--fail' in sys.argv else 0)\n")
    import socket
    for extra, want in (([], 0), (["--fail"], 1)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = b.launch_command(2, ["--steps", "1"] + extra, port)
        cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(script)
        r = subprocess.run(cmd, env=b.launch_env(2, 0, os.environ), capture_output=True, text=True, timeout=300)
        assert (r.returncode != 0) == bool(want), r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(line) == 1
        out = json.loads(line[0])
        assert out["n_gpus"] == 2 and out["sum"] == 3 and out["argv"][:2] == ["--steps", "1"]


def _ranks_worker(rank, world, port, q):
    import os
    import sys
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MPC_BENCH_BACKEND="gloo")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    r = bench.Ranks()
    try:
        ok = r.world == world and r.backend == "gloo" and r.cdev.type == "cpu"
        # 21 steps, every instance solved on every rank: the flags of 8 steps travel in one collective, the last 5 in drain()
        for _ in range(21):
            r.flag_allreduce(torch.zeros(64, dtype=torch.int32))
        r.drain()
        ok = ok and r.all_ok is True
        # one instance of rank 1 unsolved in step 13 of the next 19: every rank learns it
        for it in range(19):
            st = torch.zeros(64, dtype=torch.int32)
            if rank == 1 and it == 13:
                st[5] = 2
            r.flag_allreduce(st)
        r.drain()
        ok = ok and r.all_ok is False
        r.barrier()
        ok = ok and r.max_over_ranks(1.0 + rank) == float(world)
        q.put((rank, bool(ok)))
    finally:
        r.finish()


def test_ranks_bucketed_status_flags_two_ranks_gloo():
    """bench.py's Ranks (the N > 1 path: barrier, bucketed status all-reduce, max over ranks) with two gloo ranks on
    the CPU: what one rank fails to solve in one step reaches every rank, also from a partly filled bank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ranks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}
