"""Edge shapes and sizes of the box path against the oracle: one variable, one instance, batch sizes around the kernels'
block sizes and the small-batch switch, infinite bounds.  python tools/fuzz_edges.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd.planner import Planner
from oracle import mpc_oracle as mo
from oracle.qp_oracle import solve_box_qp
nfail = 0
rng = np.random.default_rng(1)
cases = [(1, 1, 1, 1), (1, 1, 1, 7), (2, 1, 1, 63), (1, 2, 3, 65), (3, 1, 128, 5), (12, 4, 30, 1), (12, 4, 30, 8191), (12, 4, 30, 8192),
         (12, 4, 30, 8193), (12, 4, 30, 16383), (12, 4, 30, 16385), (4, 1, 50, 24575), (4, 1, 50, 24577), (5, 3, 7, 2), (6, 2, 64, 3)]
for nx, nu, N, batch in cases:
    for inf_bounds in (False, True):
        A = 0.9 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
        B = rng.standard_normal((nx, nu)) / np.sqrt(nx) + 0.1
        Q = np.eye(nx); R = 0.1 * np.eye(nu)
        umin, umax = -0.3 * np.ones(nu), 0.4 * np.ones(nu)
        if inf_bounds:
            umin = umin.copy(); umin[0] = -np.inf; umax = umax.copy(); umax[-1] = np.inf
        x0 = rng.uniform(-2, 2, (batch, nx))
        try:
            pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 2 * Q).set_constraints(umin, umax)
            U = pl.solve(x0, want_multipliers=True).cpu().numpy().reshape(batch, -1)
        except Exception as e:      # noqa: BLE001
            print(f"nx {nx} nu {nu} N {N} batch {batch} inf {inf_bounds}: EXCEPTION {type(e).__name__}: {e}", flush=True)
            nfail += 1
            continue
        st = pl.info.status.cpu().numpy(); obj = pl.info.obj.cpu().numpy()
        H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=2 * Q))
        lb, ub = np.tile(umin, N), np.tile(umax, N)
        bad = int((st != 0).sum())
        for i in sorted(set([0, batch - 1, batch // 2] + list(rng.integers(0, batch, 3)))):
            s = solve_box_qp(H, F @ x0[i] + f0, lb, ub)
            ok = np.abs(U[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()) and \
                abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12)
            bad += int(not ok)
        print(f"nx {nx} nu {nu} N {N} batch {batch} inf {inf_bounds}: bad {bad}", flush=True)
        nfail += int(bad > 0)
print("FAILURES", nfail)
