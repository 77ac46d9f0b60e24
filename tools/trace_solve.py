import sys, os, torch
sys.path.insert(0, "/root/repo")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = int(sys.argv[1])
prob = make_config(3, batch=B, seed=0)
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
pl.opts.keep_state = 0
for _ in range(3):
    pl.solve(x0); torch.cuda.synchronize()
