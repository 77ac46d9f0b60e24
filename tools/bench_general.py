"""Config 5 timing: long-horizon MPC with state box + polytopic rows through the general-constraint path.
python tools/bench_general.py [--N 30] [--batch 256]"""
import argparse, json, sys, time
import torch
sys.path.insert(0, ".")
from mpc_amd.configs import long_horizon
from mpc_amd.planner import Planner
ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=30); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--nx", type=int, default=20); ap.add_argument("--nu", type=int, default=6)
ap.add_argument("--path", default="rows", choices=["rows", "dense"]); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
prob = long_horizon(batch=a.batch, seed=0, N=a.N, nx=a.nx, nu=a.nu)
pl = Planner(a.nx, a.nu, a.N).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.general_path = a.path
prob["x0"] = prob["x0"] * a.scale
pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"], C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
t0 = time.perf_counter(); pl.setup(); torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
t0 = time.perf_counter(); U = pl.solve(x0); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
t0 = time.perf_counter(); U = pl.solve(x0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = pl.info.status
print(json.dumps({"workload": f"long-horizon MPC nx={a.nx} nu={a.nu} N={a.N}: n={a.N*a.nu}, rows={pl.general_info()['m']}", "batch": a.batch,
                  "path": a.path, "setup_ms": t_setup * 1e3, "first_solve_ms": t_first * 1e3, "solve_ms": dt * 1e3, "solves_per_s": a.batch / dt, "solved_fraction": float((st == 0).float().mean()),
                  "active_set_steps_mean": float(pl.info.polish_steps.float().mean()), "active_set_steps_max": int(pl.info.polish_steps.max()), "status_hist": torch.bincount(st).tolist()}))
