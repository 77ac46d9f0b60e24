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


def allreduce_flag(ok, group=None) -> bool:
    """True iff every rank reports True (all-reduce MIN of one int32)."""
    t = ok if torch.is_tensor(ok) else torch.tensor([int(bool(ok))], dtype=torch.int32)
    t = t.to(torch.int32).reshape(1).clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.to(_collective_device(t, group))          # a blocking copy when it leaves the device
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


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


def solve_sharded(planner, x0_global, gather=True, group=None, **opts):
    """Every rank holds the same `x0_global` (B, nx) (or just its own rows if gather=False and it
    passes them directly); rank r solves rows shard_range(B, r, world) with its own `planner` and,
    if `gather`, all ranks end up with the full U.  Returns (U, all_solved)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x0_global.shape[0]
    lo, hi = shard_range(B, rank, world)
    U = planner.solve(x0_global[lo:hi], **opts)
    ok = bool((planner.info.status == 0).all().item())
    all_ok = allreduce_flag(torch.tensor([int(ok)], dtype=torch.int32, device=U.device), group)
    if gather:
        U = allgather_solutions(U, B, group)
    return U, all_ok
