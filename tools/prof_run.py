"""One default solve with the POLISH_PROF build (tools/prof_build/libmpcqp.so copied over the product library ON THE
GPU BOX ONLY): per-phase cycle sums of the two active-set passes.  python tools/prof_run.py [batch] [scale]"""
import os, sys
os.environ["MPCQP_DEBUG_POLISH"] = "1"
import torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sc = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
prob = make_config(3, batch=b, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"])
x0 = torch.as_tensor(prob["x0"] * sc, device="cuda:0")
pl.solve(x0); torch.cuda.synchronize()
print("---- second solve (warm caches)", file=sys.stderr)
pl.solve(x0); torch.cuda.synchronize()
