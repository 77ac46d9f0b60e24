"""Diagnostic: how well does the cold-start key (bounds violated by the unconstrained minimiser) predict the active-set work?"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = 65536
prob = make_config(3, batch=B, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"])
x0 = torch.as_tensor(prob["x0"], device="cuda:0")
U = pl.solve(x0, want_multipliers=True)
steps = pl.info.polish_steps.cpu().numpy()
nact = (pl.info.lam != 0).sum(1).cpu().numpy()
H = pl.get("H").cpu().numpy(); F = pl.get("F").cpu().numpy(); f0 = pl.get("f0").cpu().numpy()
xu = -np.linalg.solve(H, (prob["x0"] @ F.T + f0).T).T
lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
key = ((xu > ub) | (xu < lb)).sum(1)
print("corr(key, steps) = %.3f   corr(key, nact) = %.3f" % (np.corrcoef(key, steps)[0, 1], np.corrcoef(key, nact)[0, 1]))
print("key  count  steps_mean steps_max nact_mean  frac(nact>16)")
for k in sorted(set(np.minimum(key, 40))):
    m = np.minimum(key, 40) == k
    print("%3d %6d   %6.2f   %4d   %6.2f    %.3f" % (k, m.sum(), steps[m].mean(), steps[m].max(), nact[m].mean(), (nact[m] > 16).mean()))
