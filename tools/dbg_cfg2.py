import sys, numpy as np, torch
sys.path.insert(0,'.')
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
from oracle import mpc_oracle as mo
cfg,batch,scale=2,96,8.0
prob=make_config(cfg,batch=batch,seed=7); prob["x0"]=prob["x0"]*scale
pl=Planner(prob["nx"],prob["nu"],prob["N"],device=torch.device("cuda:0"))
pl.set_model(prob["A"],prob["B"]).set_cost(prob["Q"],prob["R"],prob["P"]); pl.set_constraints(prob["u_min"],prob["u_max"]); pl.setup()
U=pl.solve(prob["x0"],want_multipliers=True).cpu().numpy().reshape(batch,-1)
H,F,f0=mo.condense(prob)
lb=np.tile(prob["u_min"],prob["N"]); ub=np.tile(prob["u_max"],prob["N"])
X0=prob["x0"]; G=U@H+(X0@F.T+f0)
free=(U<ub-1e-9)&(U>lb+1e-9)
sg=np.abs(X0@F.T+f0).max(axis=1,keepdims=True)
ratio=(np.abs(G)*free/sg).max(axis=1)
act=(~free).sum(1)
lam=pl.info.lam.cpu().numpy()
st=pl.info.polish_steps.cpu().numpy()
print("inv_resid/rho", pl.get("rho") if hasattr(pl,'get') else None)
for i in np.argsort(-ratio)[:12]:
    print(i, "ratio %.2e"%ratio[i], "active",act[i], "lam_nz",(lam[i]!=0).sum(), "steps",st[i], "status",int(pl.info.status[i]))
print("bad count",(ratio>1e-8).sum(), "of",batch, " act hist", np.bincount(act))
