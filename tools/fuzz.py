"""Randomised checks of the box-QP paths against the device-side KKT certificate (shared model, batches above and below the
small-batch switch) and against the oracle (one model per instance).  python tools/fuzz.py [first_seed] [count]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.planner import Planner

def kkt(pl, x0, U, umin, umax, N, model=None, tol=1e-8):
    H, F, f0 = (pl.get(k, model=model or 0) for k in ("H", "F", "f0"))
    q = x0 @ F.t() + f0
    G = U @ H + q
    lb = torch.as_tensor(np.tile(umin, N), device="cuda:0"); ub = torch.as_tensor(np.tile(umax, N), device="cuda:0")
    scale = q.abs().max(dim=1, keepdim=True).values.clamp_min(1.0)
    free = (U < ub - 1e-9) & (U > lb + 1e-9)
    bad = (U > ub + 1e-12) | (U < lb - 1e-12) | ((G.abs() * free) > tol * scale) | ((G > tol * scale) & ~(U < ub - 1e-9) & (U > lb + 1e-9)) \
        | ((G < -tol * scale) & ~(U > lb + 1e-9) & (U < ub - 1e-9))
    bad = bad | ((G > tol * scale) & (U >= ub - 1e-9)) & False
    return bad.any(1)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfail = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    nu = int(rng.integers(1, 5)); nx = int(rng.integers(2, 13)); N = int(rng.integers(3, max(4, 128 // nu) + 1))
    batch = int(rng.choice([300, 5000, 9000, 20000]))
    A = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= min(1.0, 1.02 / np.abs(np.linalg.eigvals(A)).max())
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.02, 0.5, nu))
    w = float(rng.choice([0.02, 0.1, 0.5])); sc = float(rng.choice([0.3, 1.0, 3.0]))
    umin, umax = -w * rng.uniform(0.5, 1.5, nu), w * rng.uniform(0.5, 1.5, nu)
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 3 * Q).set_constraints(umin, umax)
    x0 = torch.as_tensor(rng.uniform(-1, 1, (batch, nx)) * sc, device="cuda:0")
    U = pl.solve(x0).reshape(batch, -1)
    st = pl.info.status
    bad = kkt(pl, x0, U, umin, umax, N) & (st == 0)
    nact = ((U >= torch.as_tensor(np.tile(umax, N), device="cuda:0") - 1e-9) | (U <= torch.as_tensor(np.tile(umin, N), device="cuda:0") + 1e-9)).sum(1)
    q = pl.get("queues").cpu().numpy().astype(int)[:4].tolist()
    msg = f"seed {seed}: nx {nx} nu {nu} N {N} n {N*nu} batch {batch} w {w} sc {sc}: status!=0 {int((st != 0).sum())}, KKT-bad among solved {int(bad.sum())}, nact mean {float(nact.float().mean()):.1f} max {int(nact.max())}, queues {q}"
    print(msg, flush=True)
    if int(bad.sum()) or int((st > 1).sum()):
        nfail += 1
    # warm start from the kept state on perturbed initial states, and after a receding-horizon shift
    x1 = x0 + 0.05 * sc * torch.as_tensor(rng.standard_normal((batch, nx)), device="cuda:0")
    U1 = pl.solve(x1, warm_start=True).reshape(batch, -1)
    bad1 = kkt(pl, x1, U1, umin, umax, N) & (pl.info.status == 0)
    pl.shift()
    U2 = pl.solve(x1, warm_start=True).reshape(batch, -1)
    bad2 = kkt(pl, x1, U2, umin, umax, N) & (pl.info.status == 0)
    print(f"         warm: KKT-bad {int(bad1.sum())} (status!=0 {int((pl.info.status != 0).sum())}), after shift {int(bad2.sum())}, steps mean {float(pl.info.polish_steps.float().mean()):.2f}", flush=True)
    if int(bad1.sum()) or int(bad2.sum()):
        nfail += 1
print("FAILURES", nfail)
