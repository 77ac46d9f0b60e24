"""In-kernel cycle counters of the split passes (needs mpc_amd/libmpcqp_prof.so: the library built with -DPOLISH_PROF)."""
import sys, os
os.environ["MPCQP_DEBUG_POLISH"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmpcqp_prof.so")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prob = make_config(3, batch=B, seed=0)
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
pl.reuse_outputs = True
for _ in range(3):
    pl.solve(x0)
torch.cuda.synchronize()
