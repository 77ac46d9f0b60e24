"""Split vs unsplit on config 3: agreement and time.  python tools/exp_split.py [batch] [x0 scale]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prob = make_config(3, batch=B, seed=0)
x0 = torch.as_tensor(prob["x0"] * scale, device="cuda:0")
res = {}
for split in (False, True):
    pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
    pl.split = split
    pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
    pl.opts.keep_state = keep
    pl.reuse_outputs = True
    U = pl.solve(x0, want_multipliers=True)
    torch.cuda.synchronize()
    print("split", split, pl.split_info()["components"], pl.split_info()["rows"], "status ok", (pl.info.status == 0).float().mean().item(),
          "steps mean", pl.info.polish_steps.float().mean().item(), "max", pl.info.polish_steps.max().item(), flush=True)
    res[split] = (U.clone(), pl.info.obj.clone(), pl.info.polish_steps.clone(), pl.info.lam.clone(), pl.get("z", batch=B).clone() if keep else None, pl.get("w", batch=B).clone() if keep else None)
    for _ in range(5): pl.solve(x0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): pl.solve(x0)
    e1.record(); torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per solve", flush=True)
    U2 = pl.solve(x0, want_multipliers=True); torch.cuda.synchronize()
    print("   repeat bit-identical:", bool((U2 == res[split][0]).all()), bool((pl.info.obj == res[split][1]).all()))
a, b = res[False], res[True]
print("max |U diff| rel", ((a[0] - b[0]).abs().max() / a[0].abs().max()).item())
print("max |obj diff| rel", ((a[1] - b[1]).abs() / a[1].abs().clamp_min(1e-30)).max().item())
print("steps equal", bool((a[2] == b[2]).all()), (a[2] - b[2]).abs().max().item())
print("max |lam diff|", (a[3] - b[3]).abs().max().item(), "scale", a[3].abs().max().item())
if keep:
    print("max |z diff|", (a[4] - b[4]).abs().max().item(), "max |w diff|", (a[5] - b[5]).abs().max().item())
