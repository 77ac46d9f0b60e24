"""Experiment: config 3 is four decoupled channels (thrust/z, roll/y, pitch/x, yaw).  Time the EXISTING kernels on each
channel as its own n=30 problem (65,536 instances each, and all four as one 262,144 batch of the heaviest) to estimate what
a component-split solve would cost.  GPU only."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner

prob = make_config(3, batch=65536, seed=0)
chan = {"thrust": ([2, 5], [0]), "roll": ([1, 4, 6, 9], [1]), "pitch": ([0, 3, 7, 10], [2]), "yaw": ([8, 11], [3])}
def run(name, sx, su, reps=1):
    sx, su = np.array(sx), np.array(su)
    A = prob["A"][np.ix_(sx, sx)]; B = prob["B"][np.ix_(sx, su)]
    Q = prob["Q"][np.ix_(sx, sx)]; R = prob["R"][np.ix_(su, su)]; P = prob["P"][np.ix_(sx, sx)]
    x0 = np.tile(prob["x0"][:, sx], (reps, 1))
    pl = Planner(len(sx), len(su), prob["N"], device="cuda:0")
    pl.set_model(A, B).set_cost(Q, R, P).set_constraints(prob["u_min"][su], prob["u_max"][su])
    pl.reuse_outputs = True
    x0d = torch.as_tensor(x0, device="cuda:0")
    for _ in range(5): pl.solve(x0d)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): pl.solve(x0d)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    st = pl.info.polish_steps.float()
    print(f"{name:8s} n={pl.n} B={x0.shape[0]}: {ms*1e3:7.1f} us per solve; steps mean {st.mean().item():.2f} max {int(st.max().item())}; solved {(pl.info.status==0).float().mean().item():.4f}", flush=True)
    return ms
tot = 0.0
for k, (sx, su) in chan.items():
    tot += run(k, sx, su)
print(f"sum of the four channels: {tot*1e3:.1f} us")
run("roll x4", *chan["roll"], reps=4)
