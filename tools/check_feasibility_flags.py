"""Do the general-constraint kernels ever call an infeasible instance solved?  An LP (largest common slack of all rows) decides
feasibility; both kernels' status flags are compared with it over a few loads.  python tools/check_feasibility_flags.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from scipy.optimize import linprog
from mpc_amd.configs import long_horizon
from mpc_amd.planner import Planner
from test_gpu_general import _oracle_rows          # tools/ are experiments: the restated rows come from the test module

bad = 0
for seed, (N, nx, nu), scale in [(11, (20, 10, 3), 1.5), (3, (8, 4, 2), 1.6), (7, (12, 6, 2), 1.8), (5, (20, 10, 3), 1.3)]:
    batch = 32
    prob = long_horizon(batch=batch, seed=seed, N=N, nx=nx, nu=nu)
    prob["x0"] *= scale
    G, E, h0 = _oracle_rows(prob)
    nvar = G.shape[1]
    cost = np.zeros(nvar + 1); cost[-1] = -1.0
    Aub = np.hstack([G, np.ones((G.shape[0], 1))])
    slack = np.array([linprog(cost, A_ub=Aub, b_ub=h0 - E @ prob["x0"][i], bounds=[(None, None)] * nvar + [(None, 10.0)], method="highs").x[-1]
                      for i in range(batch)])
    for path in ("rows", "dense"):
        pl = Planner(nx, nu, N).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.general_path = path
        pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"], C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])
        pl.solve(prob["x0"])
        st = pl.info.status.cpu().numpy()
        wrong_solved = np.where((slack < -1e-6) & (st == 0))[0]
        wrong_failed = np.where((slack > 1e-6) & (st != 0))[0]
        bad += len(wrong_solved) + len(wrong_failed)
        print(f"seed {seed} {path:5s}: infeasible {int((slack < -1e-6).sum())}/{batch}, called solved though infeasible {wrong_solved.tolist()} "
              f"(slack {slack[wrong_solved].round(4).tolist()}), feasible but not solved {wrong_failed.tolist()} (slack {slack[wrong_failed].round(4).tolist()}, status {st[wrong_failed].tolist()})")
print("MISMATCHES", bad)
