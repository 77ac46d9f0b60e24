"""Repeated solves of the same batch must agree bit for bit on every path (nothing an instance computes may depend on
scheduling): shared-model cascade at three loads, ADMM first, one model per instance, state / polytopic rows through
both kernels, the literal dense QP in a batch.  python tools/check_repro.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.configs import make_config, long_horizon
from mpc_amd.planner import Planner
bad = 0
def check(tag, solve, reps=3):
    global bad
    ref = solve().clone()
    worst = 0.0; ndiff = 0
    for _ in range(reps):
        U = solve()
        d = (U - ref).abs()
        worst = max(worst, float(d.max())); ndiff = max(ndiff, int((d.reshape(d.shape[0], -1).amax(1) > 0).sum()))
    print(f"{tag}: instances that differ {ndiff}, max |dU| {worst:.1e}" + ("" if ndiff == 0 else "   <-- NOT REPRODUCIBLE"))
    bad += ndiff != 0
p = make_config(3, batch=40000, seed=1)
pl = Planner(p["nx"], p["nu"], p["N"]).set_model(p["A"], p["B"]).set_cost(p["Q"], p["R"], p["P"]).set_constraints(p["u_min"], p["u_max"]).setup()
for sc in (1.0, 1.6, 2.5):
    x = torch.as_tensor(sc * p["x0"], device="cuda:0")
    check(f"shared model, x0 x {sc}", lambda: pl.solve(x))
x = torch.as_tensor(1.5 * p["x0"][:9000], device="cuda:0")
check("shared model, 10 ADMM iterations first", lambda: pl.solve(x, max_iter=10))
check("shared model, small batch (one pass)", lambda: pl.solve(x[:5000].contiguous()))
# the same instances in another order (other neighbours in the wave): instance by instance the same bits
perm = torch.randperm(9000, device="cuda:0")
for tag, kw, nb in (("cascade without the cold-start order", {}, 9000), ("10 ADMM iterations first", {"max_iter": 10}, 9000),
                    ("one pass", {}, 5000)):
    xa = x[:nb].contiguous(); pm = perm[perm < nb]
    Ua = pl.solve(xa, **kw).clone(); Ub = pl.solve(xa[pm].contiguous(), **kw)
    d = (Ub - Ua[pm]).abs().reshape(nb, -1).amax(1); nd = int((d > 0).sum())
    print(f"permuted batch, {tag}: instances that differ {nd}, max |dU| {float(d.max()):.1e}" + ("" if nd == 0 else "   <-- DEPENDS ON THE NEIGHBOURS"))
    bad += nd != 0
rng = np.random.default_rng(3)
Bn, N, nx, nu = 3000, 12, 4, 2
A0 = np.eye(nx) + 0.1 * np.diag(np.ones(nx - 1), 1); B0 = np.array([[0.0, 0.0], [0.005, 0.0], [0.0, 0.02], [0.1, 0.1]])
A = A0 + 0.05 * rng.standard_normal((Bn, N, nx, nx)); Bm = B0 + 0.02 * rng.standard_normal((Bn, N, nx, nu))
Q = np.diag([1.0, 2.0, 0.5, 0.1]); R = np.diag([0.05, 0.2])
pli = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, 3.0 * Q).set_constraints(-0.4 * np.ones(nu), 0.4 * np.ones(nu))
xi = torch.as_tensor(rng.standard_normal((Bn, nx)), device="cuda:0")
check("one model per instance", lambda: pli.solve(xi))
for path in ("rows", "dense"):
    g = long_horizon(batch=64, seed=3, N=10, nx=4, nu=2); g["x0"] *= 1.6
    plg = Planner(4, 2, 10).set_model(g["A"], g["B"]).set_cost(g["Q"], g["R"], g["P"]); plg.general_path = path
    plg.set_constraints(g["u_min"], g["u_max"], x_min=g["x_min"], x_max=g["x_max"], C=g["C"], c_min=g["c_min"], c_max=g["c_max"])
    xg = torch.as_tensor(g["x0"], device="cuda:0")
    check(f"state / polytopic rows, {path} kernel", lambda: plg.solve(xg))
print("NOT REPRODUCIBLE:", bad)
