"""Two ranks on ONE GPU (gloo for the collectives, host tensors) drive mpc_amd.distributed.solve_sharded with the real
Planner: shards of shard_range, all-gather of U, all-reduce of the all-solved flag -- compared with the same batch solved
by one process and with the oracle.  RCCL itself needs one GPU per rank and runs only on the driver's multi-GPU node."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpc_amd.configs import make_config
        from mpc_amd.distributed import shard_range, solve_sharded
        from mpc_amd.planner import Planner
        prob = make_config(3, batch=B, seed=5)
        pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
        pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.set_constraints(prob["u_min"], prob["u_max"])
        x0 = torch.as_tensor(prob["x0"], device="cuda:0")
        U, ok = solve_sharded(pl, x0)
        lo, hi = shard_range(B, rank, world)
        mine = pl.info.status.shape[0] == hi - lo
        U_local, ok2 = solve_sharded(pl, x0, gather=False)
        same = torch.equal(U_local, U[lo:hi])
        q.put((rank, bool(ok and ok2 and mine and same), U.cpu().numpy() if rank == 0 else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4096 + 37])
def test_two_ranks_one_gpu_real_planner(B):
    import torch
    import torch.multiprocessing as mp
    from mpc_amd.configs import make_config
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    U2 = res[0][2].reshape(B, -1)
    # one process, whole batch
    prob = make_config(3, batch=B, seed=5)
    pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
    pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"])
    U1 = pl.solve(torch.as_tensor(prob["x0"], device="cuda:0")).cpu().numpy().reshape(B, -1)
    assert np.abs(U2 - U1).max() <= 1e-11 * np.abs(U1).max()
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    for i in (0, B // 2 - 1, B // 2, B - 1):        # both sides of the shard boundary
        sol = solve_box_qp(H, F @ prob["x0"][i] + f0, lb, ub)
        assert np.abs(U2[i] - sol["x"]).max() <= 1e-6 * np.abs(sol["x"]).max()
