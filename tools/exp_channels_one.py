"""One channel of config 3 as its own n=30 problem, for profiling (see exp_channels.py): python tools/exp_channels_one.py roll 4"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
prob = make_config(3, batch=65536, seed=0)
chan = {"thrust": ([2, 5], [0]), "roll": ([1, 4, 6, 9], [1]), "pitch": ([0, 3, 7, 10], [2]), "yaw": ([8, 11], [3])}
sx, su = chan[sys.argv[1]]; reps = int(sys.argv[2])
sx, su = np.array(sx), np.array(su)
pl = Planner(len(sx), len(su), prob["N"], device="cuda:0")
pl.set_model(prob["A"][np.ix_(sx, sx)], prob["B"][np.ix_(sx, su)]).set_cost(prob["Q"][np.ix_(sx, sx)], prob["R"][np.ix_(su, su)], prob["P"][np.ix_(sx, sx)])
pl.set_constraints(prob["u_min"][su], prob["u_max"][su])
pl.reuse_outputs = True
x0d = torch.as_tensor(np.tile(prob["x0"][:, sx], (reps, 1)), device="cuda:0")
for _ in range(20): pl.solve(x0d)
torch.cuda.synchronize()
