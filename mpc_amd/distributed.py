"""Partitioning of independent MPC instances over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no counterpart: it solves one QP at a time in one process
(SE3Planner.py:108-195 couples QPs only sequentially in time).  Instances never exchange data
while being solved, so the only collectives are the two SURVEY.md section 8(e) names:
all-gather of the solutions and all-reduce of the convergence flag -- both optional for a caller
that consumes its shard locally.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world: int):
    """Contiguous row block of rank `rank`: rows [lo, hi).  Sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _collective_device(t: torch.Tensor, group=None) -> torch.device:
    """RCCL ("nccl") moves device tensors; every other backend (gloo: CPU tests, and the rehearsal of the multi-rank
    path on a box with fewer GPUs than ranks) gets host tensors."""
    if dist.get_backend(group) == "nccl":
        return t.device
    return torch.device("cpu")


class FlagReduction:
    """An all-reduce of the convergence flag in flight.  Nothing here waits for the device or for the collective until
    ``wait()`` is called: the worst status of the rank's shard is reduced ON THE DEVICE into the collective's own
    buffer (no ``.item()`` in front of the collective), and the collective is launched with ``async_op=True``."""

    def __init__(self, buf, work):
        self._buf, self._work, self._result = buf, work, None

    def wait(self) -> bool:
        """True iff every instance of every rank was solved (status 0 everywhere)."""
        if self._result is None:
            if self._work is not None:
                self._work.wait()
            self._result = int(self._buf.item()) == 0
        return self._result


def allreduce_flag_async(status, group=None) -> FlagReduction:
    """status: per-instance status codes (0 = solved) of this rank's shard, or a Python bool / 0-d tensor "all solved".
    All-reduce MAX of the worst code (one int32, latency bound); returns at once."""
    if torch.is_tensor(status):
        worst = status.reshape(-1).amax().to(torch.int32).reshape(1) if status.numel() else torch.zeros(1, dtype=torch.int32, device=status.device)
    else:
        worst = torch.tensor([0 if bool(status) else 1], dtype=torch.int32)
    work = None
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        cdev = _collective_device(worst, group)
        if cdev != worst.device:
            worst = worst.to(cdev)                      # gloo (tests, rehearsal): a blocking copy; RCCL: stays on the device
        work = dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group, async_op=True)
    return FlagReduction(worst, work)


def allreduce_flag(ok, group=None) -> bool:
    """True iff every rank reports True (blocking form of allreduce_flag_async)."""
    if torch.is_tensor(ok):
        ok = (ok.reshape(-1) == 0).logical_not().to(torch.int32) if ok.dtype == torch.bool else (ok.reshape(-1) == 0).to(torch.int32)
        return allreduce_flag_async(ok, group).wait()
    return allreduce_flag_async(bool(ok), group).wait()


def allgather_solutions(U_local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """Gather the per-rank solution blocks (rows of shard_range) into the full (batch, ...) tensor
    on every rank.  Equal shards use one all_gather_into_tensor (a direct write of each shard to
    its 7 xGMI peers); ragged shards are padded to the largest."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return U_local
    world = dist.get_world_size(group)
    home = U_local.device
    cdev = _collective_device(U_local, group)
    if cdev != home:
        return allgather_solutions_on(U_local.to(cdev), batch, group).to(home)
    return allgather_solutions_on(U_local, batch, group)


def allgather_solutions_on(U_local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """allgather_solutions on the device the tensor already lives on."""
    world = dist.get_world_size(group)
    sizes = [shard_range(batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    tail = tuple(U_local.shape[1:])
    if all(hi - lo == mx for lo, hi in sizes):
        out = torch.empty((batch,) + tail, dtype=U_local.dtype, device=U_local.device)
        dist.all_gather_into_tensor(out, U_local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tail, dtype=U_local.dtype, device=U_local.device)
    pad[: U_local.shape[0]] = U_local
    buf = torch.empty((world * mx,) + tail, dtype=U_local.dtype, device=U_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def solve_sharded(planner, x0_global, gather=True, group=None, wait=True, **opts):
    """Every rank holds the same `x0_global` (B, nx) (or just its own rows if gather=False and it
    passes them directly); rank r solves rows shard_range(B, r, world) with its own `planner` and,
    if `gather`, all ranks end up with the full U.  Returns (U, all_solved); with wait=False the second
    value is the FlagReduction still in flight (``.wait()`` it when the answer is needed), and the
    host has not synchronised with the device anywhere in the call (RCCL backend)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x0_global.shape[0]
    lo, hi = shard_range(B, rank, world)
    U = planner.solve(x0_global[lo:hi], **opts)
    flag = allreduce_flag_async(planner.info.status, group)     # device-side reduction of the shard, collective in flight
    if gather:
        U = allgather_solutions(U, B, group)
    return U, (flag.wait() if wait else flag)
