"""Randomised check of the general dense kernel (mpc_amd.solvers.qp_batch) and of the constraint-space kernel
(mpcqp_rows_*) against the oracle on small feasible QPs, semidefinite Hessians and equalities included.
python tools/fuzz_dense.py [first_seed] [count]"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
from mpc_amd import solvers, _lib
from oracle.qp_oracle import solve_qp
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load(); dev = torch.device("cuda:0")
nfail = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(2, 40)); m = int(rng.integers(1, 70)); p = int(rng.integers(0, min(4, n - 1) + 1)); B = 5
    rank = n if rng.random() < 0.6 else int(rng.integers(max(1, n // 2), n))      # sometimes semidefinite
    M = rng.standard_normal((rank, n)); P = M.T @ M / rank + (1e-3 * np.eye(n) if rank == n else 0)
    G = rng.standard_normal((m, n)); A = rng.standard_normal((p, n)) if p else None
    # bounded feasible set: a box row pair per variable keeps a semidefinite problem bounded
    G = np.vstack([G, np.eye(n), -np.eye(n)]); mm = G.shape[0]
    xf = rng.standard_normal((B, n)) * 0.5
    h = (G @ xf.T).T + rng.uniform(0.05, 1.0, (B, mm)); h[:, m:] = np.abs(xf).max() + 2.0
    b = (A @ xf.T).T if p else None
    q = rng.standard_normal((B, n)) * 2.0
    r = solvers.qp_batch(P, q, G, h, A, b)
    x = r["x"].cpu().numpy(); st = r["status"].cpu().numpy(); ob = r["obj"].cpu().numpy()
    bad = 0; worst = 0.0
    for i in range(B):
        s = solve_qp(P, q[i], G, h[i], A, None if b is None else b[i])
        if s["status"] != "optimal":
            continue
        eo = abs(ob[i] - s["primal objective"]) / max(abs(s["primal objective"]), 1e-9)
        feas = (G @ x[i] - h[i]).max(); eq = np.abs(A @ x[i] - b[i]).max() if p else 0.0
        worst = max(worst, eo)
        if st[i] != 0 or eo > 1e-7 or feas > 1e-8 or eq > 1e-8:
            bad += 1
    # constraint-space kernel on the same rows when P is definite and there are no equalities
    rows_bad = -1
    if p == 0 and rank == n:
        t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=dev).contiguous()
        per = C.c_long(0); nsh = lib.mpcqp_rows_workspace(n, mm, C.byref(per))
        sh = torch.zeros(nsh, dtype=torch.float64, device=dev); iw = torch.zeros(n, dtype=torch.int32, device=dev)
        inst = torch.zeros(B * per.value, dtype=torch.float64, device=dev)
        Pt, Gt, qt, ht = t(P), t(G), t(q), t(h)
        xr = torch.zeros((B, n), dtype=torch.float64, device=dev); zr = torch.zeros((B, mm), dtype=torch.float64, device=dev)
        obr = torch.zeros(B, dtype=torch.float64, device=dev); sr = torch.full((B,), -1, dtype=torch.int32, device=dev)
        it = torch.zeros(B, dtype=torch.int32, device=dev)
        ptr = lambda a: C.c_void_p(a.data_ptr()); stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        assert lib.mpcqp_rows_setup(0, n, mm, ptr(Pt), ptr(Gt), ptr(sh), ptr(iw), stream) == 0
        assert lib.mpcqp_rows_solve(0, B, n, mm, ptr(qt), ptr(ht), ptr(xr), ptr(zr), ptr(obr), ptr(sr), ptr(it), ptr(sh), ptr(inst), stream) == 0
        torch.cuda.synchronize()
        rows_bad = int(((sr.cpu().numpy() != 0) | (np.abs(obr.cpu().numpy() - ob) > 1e-7 * np.maximum(np.abs(ob), 1e-9))
                        | (np.abs(xr.cpu().numpy() - x).max(1) > 1e-6 * max(1.0, np.abs(x).max()))).sum())
    print(f"seed {seed}: n {n} m {mm} p {p} rank {rank}: dense bad {bad} worst obj err {worst:.1e}; rows kernel bad {rows_bad}", flush=True)
    if bad or rows_bad > 0:
        nfail += 1
print("FAILURES", nfail)
