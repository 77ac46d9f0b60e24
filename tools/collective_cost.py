"""What the per-step status all-reduce of bench.py costs, with ONE rank on RCCL (init_process_group("nccl") works with a
single rank): ms per step of (a) the solve alone, (b) + the amax reduction kernel, (c) + the asynchronous all-reduce.
RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/collective_cost.py [batch]"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", device_id=dev)
prob = make_config(3, batch=B, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
pl.reuse_outputs = True
x0 = torch.as_tensor(prob["x0"], device=dev)
flags = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(8)]
pend = [None] * 8
def run(mode, n=200):
    for it in range(n):
        pl.solve(x0)
        j = it & 7
        if mode >= 1:
            if pend[j] is not None: pend[j].wait(); pend[j] = None
            torch.amax(pl.info.status, dim=0, keepdim=True, out=flags[j])
        if mode >= 2:
            pend[j] = dist.all_reduce(flags[j], op=dist.ReduceOp.MAX, async_op=True)
    for j in range(8):
        if pend[j] is not None: pend[j].wait(); pend[j] = None
banks = [torch.zeros(8, dtype=torch.int32, device=dev) for _ in range(2)]
bp = [None, None]
def run_banks(n=200, view=True):
    tmp = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(n):
        pl.solve(x0)
        bank, j = (it >> 3) & 1, it & 7
        if j == 0 and bp[bank] is not None: bp[bank].wait(); bp[bank] = None
        if view: torch.amax(pl.info.status, dim=0, keepdim=True, out=banks[bank][j:j + 1])
        else: torch.amax(pl.info.status, dim=0, keepdim=True, out=tmp)
        if j == 7: bp[bank] = dist.all_reduce(banks[bank], op=dist.ReduceOp.MAX, async_op=True)
    for b in range(2):
        if bp[b] is not None: bp[b].wait(); bp[b] = None
for view in (True, False):
    run_banks(24, view); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run_banks(200, view); e1.record(); torch.cuda.synchronize()
    print(f"B={B} banks of 8, amax into {'a slice of the bank' if view else 'a scratch int'}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per step")
for mode, name in ((0, "solve only"), (1, "+ amax"), (2, "+ amax + async all-reduce"), (0, "solve only again")):
    run(mode, 20); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(mode); e1.record(); torch.cuda.synchronize()
    print(f"B={B} {name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per step")
dist.destroy_process_group()
