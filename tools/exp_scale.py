"""Default solve of BASELINE configs[2] with x0 scaled (more binding rows): ms per batch.  python tools/exp_scale.py [batch]"""
import sys, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prob = make_config(3, batch=B, seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"]).setup()
for scale in (1.0, 1.5, 2.5):
    x0 = torch.as_tensor(prob["x0"] * scale, device="cuda:0")
    pl.solve(x0); pl.solve(x0); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pl.solve(x0)
    e1.record(); torch.cuda.synchronize()
    q = pl.get("queues").cpu().numpy().astype(int)[:4].tolist()
    print(f"scale {scale}: {e0.elapsed_time(e1) / 10:.3f} ms  solved {int((pl.info.status == 0).sum())}/{B}  queues {q}  "
          f"steps mean {float(pl.info.polish_steps.float().mean()):.1f}")
