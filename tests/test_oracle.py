"""CPU tests: the oracle is pinned against (a) the reference scripts' own QP data, (b) a
solver-independent KKT certificate, (c) SciPy solvers that share no code with it."""
import glob
import os

import numpy as np
import pytest
import scipy.optimize as so

from oracle import mpc_oracle as mo
from oracle import reference_qps as rq
from oracle.qp_oracle import kkt_certificate, solve_box_qp, solve_qp

REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_*.npz")))


def _load(path):
    d = np.load(path)
    return [d[k] for k in "PqGhAb"], d


@pytest.mark.parametrize("path", REF, ids=[os.path.basename(p)[4:-4] for p in REF])
def test_golden_solution_is_certified(path):
    """Every committed reference QP: the stored (x, z, y) satisfies KKT to 1e-10 (scaled) -- this is
    the pin that does not depend on any solver, ours included."""
    (P, q, G, h, A, b), d = _load(path)
    c = kkt_certificate(P, q, G, h, A, b, d["x"], d["z"], d["y"])
    assert c["max"] <= 1e-10, c
    assert abs(c["objective"] - float(d["objective"])) <= 1e-12 * max(1.0, abs(c["objective"]))


@pytest.mark.parametrize("name", ["collocation1d", "launch1d", "se3_first", "launch3d"])
def test_oracle_resolves_reference_qps(name, golden_dir):
    (P, q, G, h, A, b), d = _load(os.path.join(golden_dir, f"ref_{name}.npz"))
    s = solve_qp(P, q, G, h, A if A.size else None, b if b.size else None)
    assert s["status"] == "optimal" and s["kkt"]["max"] <= 1e-10
    assert np.abs(s["x"] - d["x"]).max() <= 1e-9 * max(1.0, np.abs(d["x"]).max())


@pytest.mark.parametrize("name,fn", [("collocation1d", rq.collocation1d_qp), ("launch1d", rq.launch1d_qp),
                                     ("launch3d", rq.launch3d_qp)])
def test_restated_assembly_equals_captured_data(name, fn, golden_dir):
    """oracle/reference_qps.py reproduces what the reference script handed to solvers.qp."""
    (P, q, G, h, A, b), _ = _load(os.path.join(golden_dir, f"ref_{name}.npz"))
    got = fn()
    assert np.array_equal(got[0], P) and np.array_equal(np.ravel(got[1]), q)
    assert np.array_equal(got[2], G) and np.array_equal(np.ravel(got[3]), h)
    if A.size:
        assert np.array_equal(got[4], A) and np.array_equal(np.ravel(got[5]), b)
    else:
        assert got[4] is None


def test_cost_builder_matches_reference_formula():
    """Q = J'WJ, f = -J'Wc (HelperFuncs.py:6-11)."""
    rng = np.random.default_rng(0)
    J = rng.standard_normal((7, 4)); c = rng.standard_normal((7, 1)); W = np.diag(rng.uniform(0.1, 2, 7))
    Q, f = rq.cost_from_tasks(J, c, W)
    assert np.allclose(Q, J.T @ W @ J, rtol=0, atol=1e-14) and np.allclose(f, -(J.T @ W @ c), rtol=0, atol=1e-14)
    assert f.shape == (4, 1)


def test_oracle_vs_scipy_trust_constr():
    """Independent cross-check: SciPy trust-constr on a launch-planner QP of the reference family
    (n = 12, well scaled after a change of units) and on the 3-D launch QP objective value."""
    P, q, G, h, _, _ = rq.launch1d_qp(nodes=8, hDes=0.2)
    s = solve_qp(P, q, G, h)
    assert s["status"] == "optimal"
    sc = 1.0 / np.sqrt(np.diag(P))          # units: make the Hessian diagonal 1 for SciPy's benefit
    Ps = P * sc[:, None] * sc[None, :]; qs = q.ravel() * sc; Gs = G * sc[None, :]
    res = so.minimize(lambda x: 0.5 * x @ Ps @ x + qs @ x, np.zeros(len(qs)), jac=lambda x: Ps @ x + qs,
                      hess=lambda x: Ps, method="trust-constr",
                      constraints=[so.LinearConstraint(Gs, -np.inf, h.ravel())],
                      options=dict(gtol=1e-12, xtol=1e-14, barrier_tol=1e-12, maxiter=3000))
    # SciPy stops at its barrier tolerance (~1e-6 relative); the certified point must agree to
    # that and can only be better, never worse, while being feasible
    assert abs(res.fun - s["primal objective"]) <= 1e-5 * abs(s["primal objective"])
    assert s["primal objective"] <= res.fun + 1e-9 * abs(res.fun)
    assert (G @ s["x"] - h.ravel()).max() <= 1e-12


def test_box_oracle_vs_scipy_lsq_linear(golden_dir):
    """Box QP == bounded least squares after a Cholesky change of variables: min |L'u + L^-1 q|^2."""
    g = np.load(os.path.join(golden_dir, "mpc_cfg1.npz"))
    H, F, f0 = g["H"], g["F"], g["f0"]
    N = int(g["N"]); lb = np.tile(g["u_min"], N); ub = np.tile(g["u_max"], N)
    q = F @ g["x0"][0] + f0
    L = np.linalg.cholesky(H)
    res = so.lsq_linear(L.T, -np.linalg.solve(L, q), bounds=(lb, ub), method="bvls", tol=1e-14)
    s = solve_box_qp(H, q, lb, ub)
    assert np.abs(res.x - s["x"]).max() <= 1e-8
    assert np.abs(s["x"] - g["U"][0]).max() <= 1e-10


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_mpc_golden_certified(cfg, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mpc_cfg{cfg}.npz"))
    H, F, f0 = g["H"], g["F"], g["f0"]
    N = int(g["N"]); n = H.shape[0]
    lb = np.tile(g["u_min"], N); ub = np.tile(g["u_max"], N)
    G = np.vstack([np.eye(n), -np.eye(n)]); h = np.concatenate([ub, -lb])
    for i in range(len(g["x0"])):
        lam = g["lam"][i]
        z = np.concatenate([np.maximum(lam, 0), np.maximum(-lam, 0)])
        c = kkt_certificate(H, F @ g["x0"][i] + f0, G, h, None, None, g["U"][i], z, None)
        assert c["max"] <= 1e-10, (i, c)


def test_condense_matches_simulation():
    """X = Phi x0 + Gamma U reproduces a step-by-step simulation (LTI and LTV), and the condensed
    cost equals the stage-wise cost up to its constant."""
    rng = np.random.default_rng(1)
    nx, nu, N = 3, 2, 5
    A = rng.standard_normal((N, nx, nx)) * 0.5; B = rng.standard_normal((N, nx, nu))
    Gam, Phi = mo.rollout_matrices(A, B, N)
    x0 = rng.standard_normal(nx); U = rng.standard_normal((N, nu))
    x = x0.copy(); X = []
    for k in range(N):
        x = A[k] @ x + B[k] @ U[k]; X.append(x)
    assert np.allclose(np.concatenate(X), Phi @ x0 + Gam @ U.ravel(), atol=1e-13)
    Q = np.diag([1.0, 2.0, 3.0]); R = np.diag([0.1, 0.2]); P = 2 * Q
    prob = dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=P)
    xr = rng.standard_normal((N, nx)); ur = rng.standard_normal((N, nu))
    H, F, f0 = mo.condense(prob, xr, ur)

    def stage_cost(U):
        x = x0.copy(); c = 0.0
        for k in range(N):
            c += 0.5 * (U[k] - ur[k]) @ R @ (U[k] - ur[k])
            x = A[k] @ x + B[k] @ U[k]
            W = P if k == N - 1 else Q
            c += 0.5 * (x - xr[k]) @ W @ (x - xr[k])
        return c
    U2 = rng.standard_normal((N, nu))
    qp = lambda U: 0.5 * U.ravel() @ H @ U.ravel() + (F @ x0 + f0) @ U.ravel()
    assert abs((stage_cost(U) - stage_cost(U2)) - (qp(U) - qp(U2))) <= 1e-10


def test_admm_plus_polish_restatement_reaches_golden(golden_dir):
    """The algorithm the GPU runs (NumPy restatement) lands on the certified optimum."""
    g = np.load(os.path.join(golden_dir, "mpc_cfg3.npz"))
    N = int(g["N"]); lb = np.tile(g["u_min"], N); ub = np.tile(g["u_max"], N)
    fac = mo.factor(g["H"], g["F"], g["f0"], lb, ub, 2e-3)
    z, w = mo.admm_box(fac, g["x0"], 25)
    X, L, R = mo.polish_box(fac, g["x0"], mo.guess_side(fac, z, w), max_rounds=60)
    ok = R > 0
    assert ok.sum() >= 12           # plain block pivoting may cycle on a few; the GPU uses a dual active set
    U = X * fac["d"][None, :]
    err = np.abs(U - g["U"]).max(1) / np.abs(g["U"]).max(1)
    assert err[ok].max() <= 1e-8


def test_c_ipm_baseline_matches_golden(golden_dir):
    """oracle/qp_ipm.c (the timed CPU baseline) solves the same QPs to its tolerance."""
    from oracle import cpu_baseline as cb
    try:
        cb.load()
    except RuntimeError:
        pytest.skip("oracle/libqpipm.so not built")
    g = np.load(os.path.join(golden_dir, "mpc_cfg3.npz"))
    N = int(g["N"]); lb = np.tile(g["u_min"], N); ub = np.tile(g["u_max"], N)
    U, it, bad, _ = cb.solve_batch_box(g["H"], g["F"], g["f0"], lb, ub, g["x0"], nthreads=2,
                                       abstol=1e-12, reltol=1e-12, feastol=1e-12)
    assert bad == 0
    assert (np.abs(U - g["U"]).max(1) / np.abs(g["U"]).max(1)).max() <= 1e-6
    d = np.load(os.path.join(golden_dir, "ref_launch3d.npz"))
    x, z, st, _ = cb.solve_dense(d["P"], d["q"], d["G"], d["h"], abstol=1e-12, reltol=1e-12, feastol=1e-11)
    assert st == 0 and abs(0.5 * x @ d["P"] @ x + d["q"] @ x - float(d["objective"])) <= 1e-7 * abs(float(d["objective"]))


def test_c_activeset_baseline_matches_golden(golden_dir):
    """oracle/qp_activeset.c (the like-for-like CPU line: the GPU's algorithm on the host) lands on the certified optimum."""
    from oracle import cpu_baseline as cb
    try:
        cb.load()
    except RuntimeError:
        pytest.skip("oracle/libqpipm.so not built")
    for cfg in (1, 2, 3):
        g = np.load(os.path.join(golden_dir, f"mpc_cfg{cfg}.npz"))
        N = int(g["N"]); lb = np.tile(g["u_min"], N); ub = np.tile(g["u_max"], N)
        U, st, bad, _ = cb.solve_batch_box_activeset(g["H"], g["F"], g["f0"], lb, ub, g["x0"], nthreads=2)
        assert bad == 0 and (st >= 0).all()
        assert (np.abs(U - g["U"]).max(1) / np.abs(g["U"]).max(1)).max() <= 1e-8
        assert ((U >= ub - 1e-13) | (U <= lb + 1e-13)).sum(1).tolist() == (g["lam"] != 0).sum(1).tolist()
        # block by block where the Hessian splits (bench.py's same_algorithm_split line): the same point, the same steps
        U2, st2, bad2, _, nc = cb.solve_batch_box_activeset_split(g["H"], g["F"], g["f0"], lb, ub, g["x0"], nthreads=2)
        assert nc == (4 if cfg == 3 else 1) and bad2 == 0
        assert np.abs(U2 - U).max() <= 1e-10 * np.abs(U).max() and np.array_equal(st2, st)


def test_sweep_fixtures_are_certified_and_reproducible():
    """tests/golden/sweep_*.npz: QPs from the reference's sequential loops run end to end.  Every stored optimum
    carries a KKT certificate; re-solving a few reproduces it (the oracle is deterministic)."""
    import os
    import numpy as np
    from oracle.qp_oracle import solve_qp, kkt_certificate
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, picks in (("se3", (0, 5)), ("collocation2d", (3,))):
        g = np.load(os.path.join(gd, f"sweep_{name}.npz"))
        assert int(g["total_calls"]) >= 10
        for gi in range(int(g["n_groups"])):
            ok = g[f"g{gi}_optimal"]
            assert (g[f"g{gi}_kkt"][ok] <= 1e-9).all()
        P, q, G, h, A, b = (g[f"g0_{k}"] for k in "PqGhAb")
        for i in picks:
            s = solve_qp(P[i], q[i], G[i] if G.shape[1] else None, h[i] if G.shape[1] else None,
                         A[i] if A.shape[1] else None, b[i] if A.shape[1] else None)
            xo = g["g0_x"][i]
            assert np.abs(s["x"] - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())


def _enumerate_active_sets(P, q, G, h, A, b):
    """Solver-free: try EVERY subset S of the inequality rows as the active set, solve the equality-constrained QP
    on it by one linear solve of its KKT system, keep the candidates that are primal feasible with nonnegative
    multipliers.  For a convex QP each such candidate is a global minimiser (KKT is sufficient)."""
    import itertools
    n, m = q.size, h.size
    pe = 0 if A is None else A.shape[0]
    hs = max(1.0, np.abs(h).max())
    found = []
    for k in range(0, min(m, n - pe) + 1):
        for S in itertools.combinations(range(m), k):
            S = list(S)
            C = np.vstack(([A] if pe else []) + [G[S]]) if (pe or S) else np.zeros((0, n))
            c = np.concatenate(([b] if pe else []) + [h[S]]) if (pe or S) else np.zeros(0)
            r = C.shape[0]
            K = np.zeros((n + r, n + r)); K[:n, :n] = P
            if r:
                K[:n, n:] = C.T; K[n:, :n] = C
            # equilibrate: the reference's launch QP has Hessian entries of 1e-8 beside unit constraint rows
            s = 1.0 / np.sqrt(np.maximum(np.abs(K).max(1), 1e-300))
            Ks = K * s[:, None] * s[None, :]
            if np.linalg.matrix_rank(Ks, tol=1e-11) < n + r:
                continue                    # dependent rows or a direction of zero curvature: not a vertex candidate
            sol = s * np.linalg.solve(Ks, s * np.concatenate([-q, c]))
            x, mult = sol[:n], sol[n:]
            z = mult[pe:]
            if (G @ x - h).max() > 1e-9 * hs:
                continue
            if z.size and z.min() < -1e-9 * max(1.0, np.abs(z).max()):
                continue
            found.append((S, x, 0.5 * x @ P @ x + q @ x))
    return found


@pytest.mark.parametrize("name", ["launch1d", "collocation1d"])
def test_brute_force_active_set_enumeration_pins_the_golden_optimum(name, golden_dir):
    """Third, solver-free pin for the two reference QPs small enough to enumerate:
    LaunchPlannerQPFormulation.py:173 (2 variables, 8 rows: 37 candidate sets) and CollocationBasedPlanner.py:59
    (8 variables, 10 rows, 4 equalities: 386 candidate sets).  Every KKT point found by enumeration must coincide with
    the stored golden optimum in x and in objective."""
    (P, q, G, h, A, b), d = _load(os.path.join(golden_dir, f"ref_{name}.npz"))
    found = _enumerate_active_sets(P, q, G, h, A if A.size else None, b if b.size else None)
    assert found, "enumeration found no KKT point"
    xg, og = d["x"], float(d["objective"])
    for S, x, obj in found:
        assert abs(obj - og) <= 1e-9 * max(abs(og), 1e-300), (S, obj, og)
        assert np.abs(x - xg).max() <= 1e-7 * max(np.abs(xg).max(), 1e-300), S
    # the golden active set (rows with a positive multiplier) is among the enumerated ones
    act = sorted(np.nonzero(d["z"] > 1e-9 * max(1.0, np.abs(d["z"]).max()))[0].tolist())
    assert any(sorted(S) == act for S, _, _ in found) or any(set(act) <= set(S) for S, _, _ in found)


def test_enumeration_pins_the_first_se3_qp(golden_dir):
    """The reference's most-called QP shape, SE3Planner.py:181,183 (18 variables, 12 box rows on them: 4,096 candidate
    active sets): every KKT point the solver-free enumeration finds is the stored golden optimum."""
    (P, q, G, h, A, b), d = _load(os.path.join(golden_dir, "ref_se3_first.npz"))
    found = _enumerate_active_sets(P, q, G, h, None, None)
    assert found, "enumeration found no KKT point"
    xg, og = d["x"], float(d["objective"])
    for S, x, obj in found:
        assert abs(obj - og) <= 1e-9 * max(abs(og), 1e-300), (S, obj, og)
        assert np.abs(x - xg).max() <= 1e-7 * max(np.abs(xg).max(), 1e-300), S
    act = sorted(np.nonzero(d["z"] > 1e-9 * max(1.0, np.abs(d["z"]).max()))[0].tolist())
    assert any(sorted(S) == act for S, _, _ in found) or any(set(act) <= set(S) for S, _, _ in found)


@pytest.mark.parametrize("group,picks", [(0, (0, 5, 40)), (1, (0, 1, 2, 40))])
def test_enumeration_pins_se3_sweep_qps(group, picks, golden_dir):
    """The same pin on QPs from INSIDE the reference's backward sweep (SE3Planner.py:108-195): group 0 = the inequality-only
    calls (:183), group 1 = the calls with the 3 equality rows of the foot-contact branch (:181).  The stored solutions
    came from the oracle's interior-point restatement; the enumeration shares no code and no algorithm with it."""
    g = np.load(os.path.join(golden_dir, "sweep_se3.npz"))
    P, q, G, h, A, b = (g[f"g{group}_{k}"] for k in "PqGhAb")
    for i in picks:
        if not bool(g[f"g{group}_optimal"][i]):
            continue
        found = _enumerate_active_sets(P[i], q[i], G[i], h[i], A[i] if A.shape[1] else None, b[i] if A.shape[1] else None)
        assert found, (group, i)
        xo = g[f"g{group}_x"][i]
        oo = 0.5 * xo @ P[i] @ xo + q[i] @ xo
        for S, x, obj in found:
            assert abs(obj - oo) <= 1e-9 * max(abs(oo), 1e-300), (group, i, S)
            assert np.abs(x - xo).max() <= 1e-7 * max(np.abs(xo).max(), 1e-300), (group, i, S)
