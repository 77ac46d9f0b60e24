"""world_size-2 gloo test of the multi-GPU host logic (shards, all-gather, flag all-reduce) with a
stand-in planner: the collectives and the partitioning are exactly what runs on RCCL."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_amd.distributed import shard_range


def test_shard_range_partitions():
    for B in (1, 7, 64, 65536, 65537):
        for W in (1, 2, 3, 8):
            cuts = [shard_range(B, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


class _FakePlanner:
    """U = 2*x0 (+rank-independent), status 0 except where x0[:,0] < 0 on the designated rank."""
    def __init__(self, fail):
        self.fail = fail
        self.info = None

    def solve(self, x0, **kw):
        st = torch.zeros(x0.shape[0], dtype=torch.int32)
        if self.fail:
            st[0] = 2
        self.info = SimpleNamespace(status=st)
        return (2.0 * x0).reshape(x0.shape[0], 1, -1)


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpc_amd.distributed import allgather_solutions, allreduce_flag, solve_sharded
        x0 = torch.arange(B * 3, dtype=torch.float64).reshape(B, 3)
        U, ok = solve_sharded(_FakePlanner(fail=False), x0)
        good = torch.equal(U.reshape(B, 3), 2.0 * x0) and ok
        U2, ok2 = solve_sharded(_FakePlanner(fail=(rank == 1)), x0)      # one rank reports a failure
        good = good and (ok2 is False) and torch.equal(U2.reshape(B, 3), 2.0 * x0)
        U3, _ = solve_sharded(_FakePlanner(fail=False), x0, gather=False)
        lo, hi = shard_range(B, rank, world)
        good = good and U3.shape[0] == hi - lo
        good = good and allreduce_flag(True) is True
        # the asynchronous form: nothing is waited for until .wait()
        from mpc_amd.distributed import allreduce_flag_async
        U4, fl = solve_sharded(_FakePlanner(fail=(rank == 0)), x0, wait=False)
        good = good and hasattr(fl, "wait") and fl.wait() is False and fl.wait() is False
        good = good and allreduce_flag_async(torch.zeros(5, dtype=torch.int32)).wait() is True
        good = good and allreduce_flag(torch.tensor([0, 0, 3], dtype=torch.int32)) is False
        q.put((rank, bool(good)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 37])
def test_two_rank_gloo(B):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
