"""Host time of one Planner.solve call (batch 64: the device work is negligible). python tools/host_overhead.py"""
import sys, time, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
prob = make_config(3, batch=64, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
for reuse in (False, True):
    if hasattr(pl, "reuse_outputs"):
        pl.reuse_outputs = reuse
    for _ in range(50): pl.solve(x0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): pl.solve(x0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"reuse_outputs={reuse}: host {1e6*(t1-t0)/2000:.1f} us per call, with drain {1e6*(t2-t0)/2000:.1f} us")
f = torch.zeros(1, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): f.copy_((pl.info.status == 0).all().to(torch.int32).reshape(1))
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"flag via eq/all/to/copy: host {1e6*(t1-t0)/2000:.1f} us")
t0 = time.perf_counter()
for _ in range(2000): torch.amax(pl.info.status, dim=0, keepdim=True, out=f)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"flag via amax(out=): host {1e6*(t1-t0)/2000:.1f} us")
