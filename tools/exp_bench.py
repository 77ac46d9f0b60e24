"""Quick timing of the default solve at several batch sizes (experiments). python tools/exp_bench.py [batches...]"""
import sys, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config
from mpc_amd.planner import Planner
batches = [int(b) for b in sys.argv[1:]] or [65536, 8192]
prob = make_config(3, batch=max(batches), seed=0)
pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
pl.set_constraints(prob["u_min"], prob["u_max"])
X = torch.as_tensor(prob["x0"], device="cuda:0")
def timed(fn, reps=30):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for b in batches:
    x0 = X[:b].contiguous()
    tot = timed(lambda: pl.solve(x0)); o = timed(lambda: pl.solve(x0, stages=15)); a = timed(lambda: pl.solve(x0, stages=7)) - o
    pl.solve(x0); ok = int((pl.info.status == 0).sum()); q = pl.get("queues").cpu().numpy().astype(int)
    print(f"B={b}: total {tot*1e3:.1f} us  order {o*1e3:.1f}  passA {a*1e3:.1f}  rest {(tot-o-a)*1e3:.1f}  solved {ok}/{b}  q {q[:4].tolist()}  M solves/s {b/tot/1e3:.1f}")
