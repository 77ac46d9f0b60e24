"""GPU parity of the literal reference boundary solvers.qp(P, q, G, h[, A, b]) on the reference scripts'
own QPs (captured fixtures) and on seeded families, against the certified oracle."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _quiet():
    from mpc_amd import solvers
    solvers.options["show_progress"] = False
    yield

REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_*.npz")))


def _check(P, q, G, h, A, b, x_ref, obj_ref, sol):
    from oracle.qp_oracle import kkt_certificate
    assert sol["status"] == "optimal"
    x = np.array(sol["x"]).ravel()
    scale = max(np.abs(x_ref).max(), 1e-300)
    assert np.abs(x - x_ref).max() <= 1e-6 * scale
    assert abs(sol["primal objective"] - obj_ref) <= 1e-8 * max(abs(obj_ref), 1e-300)
    c = kkt_certificate(P, q, G, h, A, b, x, np.array(sol["z"]).ravel(), np.array(sol["y"]).ravel())
    assert c["max"] <= 1e-9, c


@pytest.mark.parametrize("path", REF, ids=[os.path.basename(p)[4:-4] for p in REF])
def test_reference_qps(path):
    from mpc_amd import solvers
    d = np.load(path)
    P, q, G, h, A, b = [d[k] for k in "PqGhAb"]
    args = [solvers.matrix(P, tc="d"), solvers.matrix(q.reshape(-1, 1), tc="d"),
            solvers.matrix(G, tc="d"), solvers.matrix(h.reshape(-1, 1), tc="d")]
    if A.size:
        args += [solvers.matrix(A, tc="d"), solvers.matrix(b.reshape(-1, 1), tc="d")]
    sol = solvers.qp(*args)
    _check(P, q, G, h, A if A.size else None, b if b.size else None, d["x"], float(d["objective"]), sol)
    # the result object behaves like the reference uses it
    xm = sol["x"]
    assert xm.shape == (P.shape[0], 1) and isinstance(xm[0], float)
    assert np.array(xm[0:2]).shape == (2, 1) or P.shape[0] < 2
    assert np.allclose(G.dot(xm).ravel(), G @ np.array(xm).ravel())


def test_reference_families_and_random():
    from mpc_amd import solvers
    from oracle import reference_qps as rq
    from oracle.qp_oracle import solve_qp
    cases = [rq.launch1d_qp(nodes=k, hDes=0.2) for k in (4, 8, 12, 20)] + [rq.launch3d_qp(nodes=k) for k in (6, 14)]
    rng = np.random.default_rng(0)
    for n, m, p in [(10, 20, 3), (40, 80, 10), (30, 10, 0), (100, 150, 40)]:
        M = rng.standard_normal((n, n)); P = M @ M.T / n + 1e-3 * np.eye(n)
        if n == 30:
            P = M[:, :15] @ M[:, :15].T / n                  # singular PSD
        q = rng.standard_normal(n); G = rng.standard_normal((m, n)); x0 = rng.standard_normal(n)
        h = G @ x0 + rng.uniform(0, 1, m)
        A = rng.standard_normal((p, n)) if p else None; b = A @ x0 if p else None
        if n == 30:
            G = np.vstack([G, np.eye(n), -np.eye(n)]); h = np.concatenate([h, 10 * np.ones(2 * n)])
        cases.append((P, q, G, h, A, b))
    for P, q, G, h, A, b in cases:
        q = np.ravel(q); h = np.ravel(h); b = None if b is None else np.ravel(b)
        s = solve_qp(P, q, G, h, A, b)
        assert s["status"] == "optimal"
        sol = solvers.qp(P, q.reshape(-1, 1), G, h.reshape(-1, 1), A, None if b is None else b.reshape(-1, 1))
        _check(P, q, G, h, A, b, s["x"], s["primal objective"], sol)


def test_status_and_errors():
    from mpc_amd import solvers
    P = np.eye(2); q = np.zeros(2)
    G = np.array([[1.0, 0.0], [-1.0, 0.0]]); h = np.array([-1.0, -1.0])        # x0 <= -1 and x0 >= 1
    assert solvers.qp(P, q, G, h)["status"] == "unknown"                      # infeasible: a status, not an exception
    with pytest.raises(ArithmeticError):
        solvers.qp(-np.eye(2), q, G, np.array([1.0, 1.0]))                    # not PSD
    with pytest.raises(ValueError):
        solvers.qp(np.eye(3), q)
    sol = solvers.qp(P, np.array([1.0, -2.0]))                                # unconstrained
    assert np.allclose(np.array(sol["x"]).ravel(), [-1.0, 2.0])
    assert solvers.matrix([], tc="d").shape == (0, 1)                         # SE3Planner.py:152
    solvers.options["show_progress"] = False


def test_degenerate_shapes_and_scales():
    """No rows, one variable, equalities only, duplicated rows, a rank-one and a zero Hessian inside a box, and problem
    data scaled by 1e+-8: every one the interior-point reference accepts (rank([P; A; G]) = n)."""
    from mpc_amd import solvers
    from oracle.qp_oracle import solve_qp
    rng = np.random.default_rng(3)
    n = 6
    M = rng.standard_normal((n, n)); P = M @ M.T + 0.1 * np.eye(n); q = rng.standard_normal(n)
    A = rng.standard_normal((2, n)); b = rng.standard_normal(2)
    G = rng.standard_normal((9, n)); h = np.abs(rng.standard_normal(9)) + 0.1
    box, one = np.vstack([np.eye(n), -np.eye(n)]), np.ones(2 * n)
    P1 = np.zeros((n, n)); P1[0, 0] = 1.0
    cases = [(P, q, None, None, None, None), (np.array([[2.0]]), np.array([1.0]), np.array([[1.0]]), np.array([-3.0]), None, None),
             (P, q, None, None, A, b), (P, q, np.vstack([G, G]), np.concatenate([h, h]), None, None),
             (P, q, None, None, np.vstack([A, A]), np.concatenate([b, b])), (P1, q, box, one, None, None),
             (np.zeros((n, n)), q, box, one, None, None), (1e8 * P, 1e8 * q, 1e-4 * G, 1e-4 * h, None, None),
             (1e-8 * P, 1e-8 * q, G, h, None, None)]
    for P_, q_, G_, h_, A_, b_ in cases:
        s = solve_qp(P_, q_, G_, h_, A_, b_)
        assert s["status"] == "optimal"
        col = lambda v: None if v is None else v.reshape(-1, 1)
        sol = solvers.qp(P_, col(q_), G_, col(h_), A_, col(b_))
        assert sol["status"] == "optimal"
        x = np.array(sol["x"]).ravel()
        assert np.abs(x - s["x"]).max() <= 1e-6 * max(np.abs(s["x"]).max(), 1.0)
        assert abs(sol["primal objective"] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-9)
    assert solvers.qp(P1, q.reshape(-1, 1))["status"] == "unknown"           # unbounded below: a status, not an exception


def test_batched_dense_and_cost_builder():
    import torch
    from mpc_amd.helper_funcs import getQuadProgCostFunction
    from mpc_amd.solvers import qp_batch
    from oracle.qp_oracle import solve_qp
    rng = np.random.default_rng(3)
    B, n, m = 9, 18, 12
    J = rng.standard_normal((B, 24, n)); c = rng.standard_normal((B, 24, 1)); W = np.stack([np.diag(rng.uniform(0.1, 2, 24))] * B)
    Q, f = getQuadProgCostFunction(torch.as_tensor(J, device="cuda"), torch.as_tensor(c, device="cuda"),
                                   torch.as_tensor(W, device="cuda"))
    Qn, fn = getQuadProgCostFunction(J[0], c[0], W[0])
    assert fn.shape == (n, 1) and np.allclose(Q[0].cpu().numpy(), Qn) and np.allclose(f[0].cpu().numpy(), fn)
    G = rng.standard_normal((B, m, n)); h = rng.uniform(0.1, 1, (B, m))
    r = qp_batch(Q, f[:, :, 0], G, h)
    assert (r["status"].cpu().numpy() == 0).all()
    X = r["x"].cpu().numpy()
    for i in range(B):
        s = solve_qp(Q[i].cpu().numpy(), f[i, :, 0].cpu().numpy(), G[i], h[i])
        assert np.abs(X[i] - s["x"]).max() <= 1e-6 * max(np.abs(s["x"]).max(), 1e-300)


@pytest.mark.parametrize("name", ["se3", "collocation2d"])
def test_reference_sequential_sweeps_replay(name, golden_dir):
    """QPs the reference's sequential loops build from the answers to their previous QPs
    (SE3Planner.py:103-223 backward sweeps, CollocationBasedPlanner2D.py:85-365 SQP iterations), captured by running
    the scripts end to end (tests/golden/make_golden.py sweep) -- solved here as batches through the C-ABI."""
    import torch
    from mpc_amd import solvers
    g = np.load(os.path.join(golden_dir, f"sweep_{name}.npz"))
    checked = 0
    for gi in range(int(g["n_groups"])):
        P, q, G, h, A, b = (g[f"g{gi}_{k}"] for k in "PqGhAb")
        r = solvers.qp_batch(P, q, G if G.shape[1] else None, h if G.shape[1] else None,
                             A if A.shape[1] else None, b if A.shape[1] else None)
        x = r["x"].cpu().numpy(); obj = r["obj"].cpu().numpy(); st = r["status"].cpu().numpy()
        for i in np.nonzero(g[f"g{gi}_optimal"])[0]:
            xo, oo = g[f"g{gi}_x"][i], g[f"g{gi}_obj"][i]
            if not np.isfinite(xo).all() or np.abs(xo).max() > 1e8:
                continue            # the script's own iteration has blown up by then (SURVEY.md appendix, SE3Planner.py:114)
            assert st[i] == 0, (gi, i)
            assert np.abs(x[i] - xo).max() <= 1e-6 * max(1.0, np.abs(xo).max()), (gi, i)
            assert abs(obj[i] - oo) <= 1e-8 * max(1.0, abs(oo)), (gi, i)
            checked += 1
    # every call of the script whose own iterate is still finite (SE3Planner.py's rollout diverges after ~95 QPs, SURVEY.md
    # appendix) and all 10 SQP iterations of CollocationBasedPlanner2D.py
    assert checked == {"se3": 95, "collocation2d": 10}[name]


def test_remove_null_constraints_then_qp(golden_dir, capsys):
    """HelperFuncs.py:41-54 on the device, against what the reference's own function returned (fixture written by
    tests/golden/make_golden.py from /root/reference), then the pruned rows through solvers.qp as
    CollocationBasedPlanner2D.py:338-359 does."""
    import torch
    from mpc_amd import solvers
    from mpc_amd.helper_funcs import removeNullConstraints
    g = np.load(os.path.join(golden_dir, "null_rows.npz"))
    thr = float(g["threshold"])
    for tag in ("plain", "injected"):
        J, c = g[f"{tag}_J"], g[f"{tag}_c"]
        capsys.readouterr()
        Jo, co = removeNullConstraints(J, c, thr)
        text = capsys.readouterr().out
        assert isinstance(Jo, np.ndarray) and co.shape == (Jo.shape[0], 1)
        assert np.array_equal(Jo, g[f"{tag}_J_out"]) and np.array_equal(co, g[f"{tag}_c_out"])          # bit-exact, same order
        assert text.count("non zero objective") == int(g[f"{tag}_warnings"])
        assert f"Removing {int(g[f'{tag}_removed'])} constraints" in text
        Jt, ct = removeNullConstraints(torch.as_tensor(J, device="cuda"), torch.as_tensor(c, device="cuda"), thr)
        assert Jt.is_cuda and np.array_equal(Jt.cpu().numpy(), Jo) and np.array_equal(ct.cpu().numpy(), co)
    # the script's step: stack, prune, solve (the injected null rows must not change the answer)
    d = np.load(os.path.join(golden_dir, "ref_collocation2d_iter0.npz"))
    Aeq, beq = removeNullConstraints(g["injected_J"], g["injected_c"], thr)
    solvers.options["show_progress"] = False
    sol = solvers.qp(solvers.matrix(d["P"], tc="d"), solvers.matrix(d["q"].reshape(-1, 1), tc="d"),
                     solvers.matrix(d["G"], tc="d"), solvers.matrix(d["h"].reshape(-1, 1), tc="d"),
                     solvers.matrix(Aeq, tc="d"), solvers.matrix(beq, tc="d"))
    assert sol["status"] == "optimal"
    x = np.array(sol["x"]).ravel()
    assert np.abs(x - d["x"]).max() <= 1e-6 * np.abs(d["x"]).max()
    assert abs(sol["primal objective"] - float(d["objective"])) <= 1e-8 * abs(float(d["objective"]))
    z0 = torch.zeros((0, 5), dtype=torch.float64, device="cuda")
    Je, ce = removeNullConstraints(z0, torch.zeros((0, 1), dtype=torch.float64, device="cuda"), thr)     # empty stack
    assert tuple(Je.shape) == (0, 5) and tuple(ce.shape) == (0, 1)


def test_options_follow_cvxopt_semantics(capsys):
    """solvers.options as the reference sets them (CollocationBasedPlanner2D.py:16-20, SE3Planner.py:8)."""
    import warnings
    from mpc_amd import solvers
    from oracle import reference_qps as rq
    P, q, G, h, _, _ = rq.launch3d_qp()
    saved = dict(solvers.options)
    try:
        solvers.options.clear()
        capsys.readouterr()
        full = solvers.qp(P, q, G, h)
        assert "mpc_amd.solvers.qp" in capsys.readouterr().out          # show_progress defaults to True, as in cvxopt
        assert full["status"] == "optimal" and full["iterations"] > 3
        assert full["gap"] <= 1e-7 and full["primal infeasibility"] <= 1e-7 and full["dual infeasibility"] <= 1e-7
        assert abs(full["dual objective"] - full["primal objective"]) <= 1e-8 * abs(full["primal objective"])
        solvers.options["show_progress"] = False
        solvers.qp(P, q, G, h)
        assert capsys.readouterr().out == ""
        solvers.options["maxiters"] = 3                                  # cap reached: a status, not an exception
        capped = solvers.qp(P, q, G, h)
        assert capped["status"] == "unknown" and capped["iterations"] <= 4
        solvers.options.update(maxiters=500, refinement=1, reltol=1e-4, abstol=1e-5, feastol=1e-5)    # Collocation2D's values
        assert solvers.qp(P, q, G, h)["status"] == "optimal"
        solvers.options.clear()
        solvers.options.update(show_progress=False, feastol=1e-300, abstol=0.0, reltol=0.0)    # a gate nothing can meet
        strict = solvers.qp(P, q, G, h)
        assert strict["status"] == "unknown"
        assert np.array_equal(np.array(strict["x"]), np.array(full["x"]))   # the tolerances gate the status only
        solvers.options["maxiters"] = 0
        with pytest.raises(ValueError):
            solvers.qp(P, q, G, h)
        solvers.options.clear(); solvers.options["show_progress"] = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            solvers.qp(P, q, G, h, initvals={"x": np.zeros(P.shape[0])})
        assert any("initvals" in str(x.message) for x in w)
    finally:
        solvers.options.clear(); solvers.options.update(saved)
    # cvxopt's nested-list convention: inner lists are columns
    m = solvers.matrix([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    assert m.shape == (2, 3) and m[1, 0] == 2.0 and m.size == (2, 3) and m[2] == 3.0
