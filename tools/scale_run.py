"""A few default solves of BASELINE configs[2] with x0 scaled (more binding rows) -- for rocprofv3 kernel stats.
python tools/scale_run.py [scale] [batch] [reps]"""
import sys, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = make_config(3, batch=B, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
x0 = torch.as_tensor(prob["x0"] * scale, device="cuda:0")
for _ in range(reps):
    pl.solve(x0)
torch.cuda.synchronize()
print("solved", int((pl.info.status == 0).sum()), "of", B, "steps mean", float(pl.info.polish_steps.float().mean()))
