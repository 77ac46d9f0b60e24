"""ms per solve of config 3 over batch sizes, split on / off: python tools/sweep_split.py [keep_state]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
keep = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prob = make_config(3, batch=65536, seed=0)
pls = {}
for split in (True, False):
    pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
    pl.split = split; pl.opts.keep_state = keep
    pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
    pl.reuse_outputs = True
    pls[split] = pl
xall = torch.as_tensor(prob["x0"], device="cuda:0")
for b in (64, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536):
    x = xall[:b].contiguous()
    out = []
    for split in (True, False):
        pl = pls[split]
        for _ in range(5): pl.solve(x)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): pl.solve(x)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 40 * 1e3)
        assert bool((pl.info.status == 0).all())
    print(f"batch {b:6d}: split {out[0]:7.1f} us   whole {out[1]:7.1f} us", flush=True)
