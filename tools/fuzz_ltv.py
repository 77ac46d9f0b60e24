"""Randomised check of the one-model-per-instance path (assembly, per-model factor, active-set solve) against the KKT
certificate evaluated on the host from the oracle's condensing.  python tools/fuzz_ltv.py [first_seed] [count]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.planner import Planner
from oracle import mpc_oracle as mo
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfail = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    nu = int(rng.integers(1, 4)); nx = int(rng.integers(2, 13)); N = int(rng.integers(3, max(4, 128 // nu) + 1))
    nb = int(rng.choice([33, 200])); tv = bool(rng.integers(0, 2))
    shp = (nb, N if tv else 1)
    A = np.eye(nx) + 0.12 * rng.standard_normal(shp + (nx, nx)) / np.sqrt(nx)
    A *= 0.97
    Bm = rng.standard_normal(shp + (nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.02, 0.5, nu)); P = 3 * Q
    w = float(rng.choice([0.05, 0.3])); sc = float(rng.choice([0.5, 2.0]))
    umin, umax = -w * rng.uniform(0.5, 1.5, nu), w * rng.uniform(0.5, 1.5, nu)
    xr = 0.3 * rng.standard_normal((nb, N, nx)); ur = 0.1 * rng.standard_normal((nb, N, nu))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P, x_ref=xr, u_ref=ur).set_constraints(umin, umax)
    x0 = rng.uniform(-1, 1, (nb, nx)) * sc
    U = pl.solve(torch.as_tensor(x0, device="cuda:0"), want_multipliers=True).cpu().numpy().reshape(nb, -1)
    st = pl.info.status.cpu().numpy(); lam = pl.info.lam.cpu().numpy(); obj = pl.info.obj.cpu().numpy()
    lb = np.tile(umin, N); ub = np.tile(umax, N)
    worst = 0.0; nbad = 0
    for i in rng.choice(nb, 6, replace=False):
        Ai = A[i] if tv else A[i, 0]; Bi = Bm[i] if tv else Bm[i, 0]
        H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=Ai, B=Bi, Q=Q, R=R, P=P), x_ref=xr[i], u_ref=ur[i])
        q = F @ x0[i] + f0
        g = H @ U[i] + q
        scale = max(1.0, np.abs(q).max())
        free = (U[i] < ub - 1e-9) & (U[i] > lb + 1e-9)
        e = max(np.abs(g[free]).max() if free.any() else 0.0, np.maximum(g, 0)[U[i] <= lb + 1e-9].max() if (U[i] <= lb + 1e-9).any() and False else 0.0)
        e = max(e, np.maximum(g[(U[i] >= ub - 1e-9)], 0).max() if (U[i] >= ub - 1e-9).any() else 0.0)     # at ub: g <= 0
        e = max(e, np.maximum(-g[(U[i] <= lb + 1e-9)], 0).max() if (U[i] <= lb + 1e-9).any() else 0.0)    # at lb: g >= 0
        feas = max((U[i] - ub).max(), (lb - U[i]).max())
        o = 0.5 * U[i] @ H @ U[i] + q @ U[i]
        eo = abs(o - obj[i]) / max(abs(o), 1e-9)
        worst = max(worst, e / scale)
        if st[i] == 0 and (e > 1e-7 * scale or feas > 1e-12 or eo > 1e-7):
            nbad += 1
    nact = ((U >= ub - 1e-9) | (U <= lb + 1e-9)).sum(1)
    print(f"seed {seed}: nx {nx} nu {nu} N {N} n {N*nu} models {nb} tv {tv}: status!=0 {int((st != 0).sum())} bad {nbad} worst KKT {worst:.1e} nact mean {nact.mean():.1f} max {nact.max()}", flush=True)
    if nbad or (st > 1).any():
        nfail += 1
print("FAILURES", nfail)
