import sys, numpy as np, torch
sys.path.insert(0,'.')
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
prob=make_config(3,batch=4096,seed=0)
pl=Planner(prob["nx"],prob["nu"],prob["N"],device=torch.device("cuda:0"))
pl.set_model(prob["A"],prob["B"]).set_cost(prob["Q"],prob["R"],prob["P"]); pl.set_constraints(prob["u_min"],prob["u_max"]); pl.setup()
x0=torch.as_tensor(prob["x0"],device="cuda:0")
U=pl.solve(x0); torch.cuda.synchronize()
print("steps mean",pl.info.polish_steps.float().mean().item(), "status0", (pl.info.status==0).float().mean().item())
U=pl.solve(x0, want_multipliers=True); torch.cuda.synchronize()
lam=pl.info.lam
act=(lam!=0).sum(1).float()
print("active at solution: mean",act.mean().item(),"max",act.max().item(),"frac>10",(act>10).float().mean().item())
ps=pl.info.polish_steps.float()
print("steps hist", torch.bincount(pl.info.polish_steps.clamp(max=30)).tolist())
for t in (8,10,12,14,16,18,20,22,24):
    print("frac k>",t,(act>t).float().mean().item())
a8=act.view(-1,8).max(1).values; a16=act.view(-1,16).max(1).values
print("mean of max over 8:",a8.mean().item()," over 16:",a16.mean().item())
st=pl.info.polish_steps.float()
print("steps: mean of max over 8:",st.view(-1,8).max(1).values.mean().item()," over 16:",st.view(-1,16).max(1).values.mean().item())
