"""Host time of one Planner.solve call: a burst of 8 calls issued into an EMPTY stream (so that the host never waits for
queue space -- a long loop measures the device's time per solve by back-pressure, not the host's).
python tools/host_overhead.py"""
import statistics, sys, time, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
prob = make_config(3, batch=64, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
for reuse in (False, True):
    pl.reuse_outputs = reuse
    for _ in range(20): pl.solve(x0)
    host, total = [], []
    for _ in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): pl.solve(x0)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) / 8); total.append((t2 - t0) / 8)
    print(f"reuse_outputs={reuse}: host {1e6*statistics.median(host):.1f} us per call; device-bound rate at batch 64: "
          f"{1e6*statistics.median(total):.1f} us per solve")
f = torch.zeros(1, dtype=torch.int32, device="cuda:0")
host = []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): torch.amax(pl.info.status, dim=0, keepdim=True, out=f)
    host.append((time.perf_counter() - t0) / 8)
print(f"flag via amax(out=): host {1e6*statistics.median(host):.1f} us")
