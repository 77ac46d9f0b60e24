"""Randomised checks of the less-travelled variants of the box path: ADMM iterations before the active-set stage, per-stage
bounds with open sides, per-instance models with per-instance bounds.  python tools/fuzz_variants.py [first_seed] [count]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.planner import Planner
from oracle import mpc_oracle as mo
from oracle.qp_oracle import solve_box_qp
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfail = 0
def check(tag, U, obj, st, Hs, qs, lb, ub, idx):
    bad = int((st != 0).sum())
    for i in idx:
        s = solve_box_qp(Hs(i), qs(i), lb(i), ub(i))
        ok = np.abs(U[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()) and \
            abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12)
        bad += int(not ok)
    print(f"   {tag}: bad {bad}", flush=True)
    return int(bad > 0)
for seed in range(first, first + count):
    rng = np.random.default_rng(4000 + seed)
    nu = int(rng.integers(1, 4)); nx = int(rng.integers(2, 10)); N = int(rng.integers(2, max(3, 100 // nu)))
    n = N * nu
    A = 0.92 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.05, 0.5, nu)); P = 2 * Q
    w = float(rng.choice([0.05, 0.3]))
    print(f"seed {seed}: nx {nx} nu {nu} N {N} n {n} w {w}", flush=True)
    H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=P))
    # (a) ADMM iterations first, shared model, per-stage bounds with open sides
    batch = int(rng.choice([100, 9000]))
    lbs = -w * rng.uniform(0.5, 1.5, (N, nu)); ubs = w * rng.uniform(0.5, 1.5, (N, nu))
    lbs[rng.integers(0, N), rng.integers(0, nu)] = -np.inf; ubs[rng.integers(0, N), rng.integers(0, nu)] = np.inf
    x0 = rng.uniform(-1, 1, (batch, nx)) * float(rng.choice([0.5, 2.0]))
    for K in (0, 3):
        pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, P).set_constraints(lbs, ubs)
        U = pl.solve(x0, max_iter=K).cpu().numpy().reshape(batch, -1)
        nfail += check(f"per-stage bounds, batch {batch}, max_iter {K}", U, pl.info.obj.cpu().numpy(), pl.info.status.cpu().numpy(),
                       lambda i: H, lambda i: F @ x0[i] + f0, lambda i: lbs.reshape(-1), lambda i: ubs.reshape(-1),
                       rng.integers(0, batch, 5))
    # (b) one model per instance with per-instance bounds (the LTV deviation form), open sides included
    nb = 40
    Ab = A + 0.05 * rng.standard_normal((nb, N, nx, nx)); Bb = B + 0.1 * rng.standard_normal((nb, N, nx, nu))
    lbi = -w * rng.uniform(0.5, 1.5, (nb, N, nu)); ubi = w * rng.uniform(0.5, 1.5, (nb, N, nu))
    lbi[rng.integers(0, nb, 5), rng.integers(0, N, 5), rng.integers(0, nu, 5)] = -np.inf
    xr = 0.2 * rng.standard_normal((nb, N, nx)); ur = 0.1 * rng.standard_normal((nb, N, nu))
    x0b = rng.uniform(-1, 1, (nb, nx))
    pl = Planner(nx, nu, N).set_model(Ab, Bb).set_cost(Q, R, P, x_ref=xr, u_ref=ur).set_constraints(lbi, ubi)
    U = pl.solve(x0b).cpu().numpy().reshape(nb, -1)
    cache = {}
    def cond(i):
        if i not in cache:
            cache[i] = mo.condense(dict(N=N, nx=nx, nu=nu, A=Ab[i], B=Bb[i], Q=Q, R=R, P=P), x_ref=xr[i], u_ref=ur[i])
        return cache[i]
    nfail += check("per-instance models and bounds", U, pl.info.obj.cpu().numpy(), pl.info.status.cpu().numpy(),
                   lambda i: cond(i)[0], lambda i: cond(i)[1] @ x0b[i] + cond(i)[2], lambda i: lbi[i].reshape(-1),
                   lambda i: ubi[i].reshape(-1), rng.integers(0, nb, 5))
print("FAILURES", nfail)
