"""Randomised check of the general-constraint MPC mode (state box + polytopic rows behind mpcqp_factor / mpcqp_solve, both
kernels, shared model and one model per instance) against the oracle.  python tools/fuzz_general.py [first_seed] [count]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.planner import Planner
from oracle import mpc_oracle as mo
from oracle.qp_oracle import solve_qp
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfail = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(3000 + seed)
    nu = int(rng.integers(1, 4)); nx = int(rng.integers(2, 8)); N = int(rng.integers(2, 14)); nc = int(rng.integers(0, 3)); nb = 6
    per = bool(rng.integers(0, 3) == 0)                      # one model per instance in a third of the cases
    shp = (nb, N) if per else ()
    A = (0.95 * np.eye(nx) + 0.1 * rng.standard_normal(shp + (nx, nx)) / np.sqrt(nx))
    Bm = rng.standard_normal(shp + (nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.05, 0.5, nu)); P = 2 * Q
    umin, umax = -rng.uniform(0.2, 1.0, nu), rng.uniform(0.2, 1.0, nu)
    xw = float(rng.choice([0.6, 1.5, 4.0]))
    xmin = -xw * rng.uniform(0.7, 1.3, nx); xmax = xw * rng.uniform(0.7, 1.3, nx)
    xmin[rng.integers(0, nx)] = -np.inf
    if rng.random() < 0.5: xmax[rng.integers(0, nx)] = np.inf
    Cm = rng.standard_normal((nc, nx)) if nc else None
    cmin = -xw * np.ones(nc) if nc else None; cmax = xw * rng.uniform(0.8, 1.2, nc) if nc else None
    x0 = rng.uniform(-1, 1, (nb, nx)) * 0.8
    for path in (("dense",) if per else ("rows", "dense")):
        pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P)
        pl.general_path = path
        pl.set_constraints(umin, umax, x_min=xmin, x_max=xmax, C=Cm, c_min=cmin, c_max=cmax)
        U = pl.solve(x0, want_multipliers=True).cpu().numpy().reshape(nb, -1)
        st = pl.info.status.cpu().numpy(); obj = pl.info.obj.cpu().numpy()
        bad = 0; nopt = 0
        for i in range(nb):
            prob = dict(N=N, nx=nx, nu=nu, A=A[i] if per else A, B=Bm[i] if per else Bm, Q=Q, R=R, P=P)
            H, F, f0 = mo.condense(prob)
            Gam, Phi = mo.rollout_matrices(prob["A"], prob["B"], N)
            n = N * nu; I = np.eye(n); Z = np.zeros((n, nx))
            rows = [(I, Z, np.tile(umax, N)), (-I, Z, -np.tile(umin, N)), (Gam, Phi, np.tile(xmax, N)), (-Gam, -Phi, -np.tile(xmin, N))]
            if nc:
                Cb = np.kron(np.eye(N), Cm)
                rows += [(Cb @ Gam, Cb @ Phi, np.tile(cmax, N)), (-Cb @ Gam, -Cb @ Phi, -np.tile(cmin, N))]
            G = np.vstack([r[0] for r in rows]); E = np.vstack([r[1] for r in rows]); h0 = np.concatenate([r[2] for r in rows])
            keep = np.isfinite(h0); G, E, h0 = G[keep], E[keep], h0[keep]
            s = solve_qp(H, F @ x0[i] + f0, G, h0 - E @ x0[i])
            if s["status"] != "optimal":
                bad += int(st[i] == 0)
                continue
            nopt += 1
            ok = st[i] == 0 and np.abs(U[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()) \
                and abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12)
            bad += int(not ok)
        print(f"seed {seed}: nx {nx} nu {nu} N {N} nc {nc} per-instance {per} path {path} m {pl.general_info()['m']}: optimal {nopt}/{nb} bad {bad}", flush=True)
        nfail += int(bad > 0)
print("FAILURES", nfail)
