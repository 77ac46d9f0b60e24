import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_split as T
for name, umin, umax in (("lo-inf", [-np.inf, -0.25, -0.2], [0.2, 0.3, 0.25]), ("hi-inf", [-0.3, -0.25, -0.2], [0.2, np.inf, 0.25]),
                         ("both", [-np.inf, -0.25, -0.2], [0.2, np.inf, 0.25]), ("none", [-0.3, -0.25, -0.2], [0.2, 0.3, 0.25])):
    for batch in (300, 20000):
        prob = T._block_problem([(3, 1), (3, 1), (4, 1)], N=16, batch=batch, seed=21, ubound=0.25, x0_amp=4.0, stab=0.9)
        prob["u_min"] = np.array(umin); prob["u_max"] = np.array(umax)
        pl = T._planner(prob); U = pl.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
        pu = T._planner(prob, split=False); Uu = pu.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
        lb = np.tile(prob["u_min"], 16); ub = np.tile(prob["u_max"], 16)
        print(name, batch, "comps", pl.split_info()["components"], "split viol", int(((U > ub + 1e-9) | (U < lb - 1e-9)).sum()),
              "unsplit viol", int(((Uu > ub + 1e-9) | (Uu < lb - 1e-9)).sum()), "steps", pl.info.polish_steps.cpu().numpy()[:6], pu.info.polish_steps.cpu().numpy()[:6],
              "status", int(pl.info.status.max()), int(pu.info.status.max()))
