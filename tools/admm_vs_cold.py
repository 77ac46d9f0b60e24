"""When do ADMM iterations before the active-set stage pay?  Solve time of BASELINE configs[2] (65,536 instances) for
several x0 scales (larger = more binding rows) and ADMM iteration counts.  python tools/admm_vs_cold.py"""
import json, sys, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = 65536
prob = make_config(3, batch=B, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
for scale in (1.0, 1.5, 2.5):
    x0 = torch.as_tensor(prob["x0"] * scale, device="cuda:0")
    row = {"x0_scale": scale}
    for K in (0, 5, 10, 20):
        pl.solve(x0, max_iter=K, want_multipliers=True); torch.cuda.synchronize()
        nact = float((pl.info.lam != 0).sum(1).float().mean())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            pl.solve(x0, max_iter=K)
        e1.record(); torch.cuda.synchronize()
        row[f"ms_K{K}"] = round(e0.elapsed_time(e1) / 10, 3)
        row["active_rows_mean"] = round(nact, 1)
        row[f"solved_K{K}"] = float((pl.info.status == 0).float().mean())
    print(json.dumps(row))
