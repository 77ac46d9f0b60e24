"""ms per default solve of BASELINE configs[2] over batch sizes (looks for cliffs between the size classes of the cascade).
python tools/batch_sweep.py"""
import sys, time, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
p = make_config(3, batch=65536, seed=0)
pl = Planner(p["nx"], p["nu"], p["N"]).set_model(p["A"], p["B"]).set_cost(p["Q"], p["R"], p["P"]).set_constraints(p["u_min"], p["u_max"]).setup()
pl.reuse_outputs = True
x = torch.as_tensor(p["x0"], device="cuda:0")
for nb in (64, 1024, 4096, 8192, 8193, 12288, 16384, 16385, 20000, 24575, 24576, 28000, 32768, 40000, 49152, 57344, 65536):
    xs = x[:nb].contiguous()
    for _ in range(5): pl.solve(xs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): pl.solve(xs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print(f"{nb:6d}  {dt * 1e3:.4f} ms  {nb / dt / 1e6:7.1f} M solves/s  solved {int((pl.info.status == 0).sum())}")
