"""Host wall time of the set-up of one shared model (mpcqp_assemble + mpcqp_factor, the synchronous call that also looks
for the component split), repeated on a warm handle: python tools/setup_time.py [config] [repeats]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prob = make_config(cfg, batch=64, seed=0)
for split in (True, False):
    pl = Planner(prob["nx"], prob["nu"], prob["N"], device="cuda:0")
    pl.split = split
    pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
    pl.setup(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        pl.set_cost(prob["Q"], prob["R"], prob["P"])          # marks the handle dirty: the next setup() assembles and factors again
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.setup(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"config {cfg} split {split}: components {pl.split_info()['components']}, set-up median {ts[len(ts) // 2]:.2f} ms, min {ts[0]:.2f} ms")
