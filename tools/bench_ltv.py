"""Config 4 timing: kinematic-bicycle LTV MPC closed loop, 8,192 parallel rollouts, warm-start shifting.
python tools/bench_ltv.py [--batch 8192] [--steps 200]"""
import argparse, json, sys, time
import torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.ltv import Bicycle, ClosedLoopLTV

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--cold", action="store_true")
ap.add_argument("--graph", action="store_true", help="replay the step as one HIP graph")
ap.add_argument("--no-stats", action="store_true", help="do not read per-step statistics back (no host sync per step)")
a = ap.parse_args()
cfg = make_config(4, batch=a.batch, seed=0)
dev = torch.device("cuda:0")
N, dt = cfg["N"], cfg["dt"]
loop = ClosedLoopLTV(Bicycle(cfg["L"], dt), N, cfg["Q"], cfg["R"], cfg["P"], cfg["u_min"], cfg["u_max"], device=dev)
x = torch.as_tensor(cfg["x0"], device=dev)
ks = torch.arange(1, N + 1, device=dev, dtype=torch.float64)
def one(x):
    x_ref = torch.zeros((a.batch, N, 4), dtype=torch.float64, device=dev)
    x_ref[:, :, 0] = x[:, :1] + cfg["v_ref"] * dt * ks
    x_ref[:, :, 3] = cfg["v_ref"]
    if graph_on:
        u0 = loop.replay(x, x_ref)
    else:
        u0 = loop.step(x, x_ref, warm=not a.cold)
    return loop.model.step(x, u0), x_ref
graph_on = False
for _ in range(5):
    x, xr = one(x)
if a.graph:
    loop.capture(x, xr, warm=not a.cold)
    graph_on = True
    for _ in range(3):
        x, xr = one(x)
torch.cuda.synchronize()
t0 = time.perf_counter(); st = 0.0; solved = 0
for _ in range(a.steps):
    x, xr = one(x)
    if not a.no_stats:
        st += float(loop.pl.info.polish_steps.float().mean()); solved += int((loop.pl.info.status == 0).sum())
torch.cuda.synchronize()
dt_ = time.perf_counter() - t0
if a.no_stats:
    st = float(loop.pl.info.polish_steps.float().mean()) * a.steps; solved = int((loop.pl.info.status == 0).sum()) * a.steps
print(json.dumps({"workload": "configs[3]: kinematic bicycle LTV closed loop", "rollouts": a.batch, "steps": a.steps,
                  "graph": bool(a.graph), "per_step_stats": not a.no_stats,
                  "ms_per_closed_loop_step": dt_ / a.steps * 1e3, "qp_solves_per_s": a.batch * a.steps / dt_,
                  "active_set_steps_mean": st / a.steps, "solved_fraction": solved / (a.batch * a.steps),
                  "final_abs_y_max": float(x[:, 1].abs().max())}))
