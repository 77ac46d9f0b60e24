"""Queue sizes and key histogram of the default solve (mpcqp_get selector 12). python tools/queue_stats.py [--batch B] [--scale s]"""
import argparse, sys
import torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=65536); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
prob = make_config(3, batch=a.batch, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"])
x0 = torch.as_tensor(prob["x0"] * a.scale, device="cuda:0")
pl.solve(x0)
q = pl.get("queues").cpu().numpy().astype(int)
st = pl.info.polish_steps.cpu().numpy()
print(f"batch {a.batch} scale {a.scale}: second pass {q[0]} (first pass took {q[2]}), 64-row pass {q[1]}, any-size pass {q[3]}")
print("key histogram:", q[16:48].tolist())
print("steps mean %.2f max %d; solved %d" % (st.mean(), st.max(), int((pl.info.status == 0).sum())))
