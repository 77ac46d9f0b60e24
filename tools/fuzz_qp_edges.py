"""Edge cases of the literal solvers.qp call: no constraints, one variable, equalities only, infeasible rows, a direction
of unbounded descent, duplicate rows.  python tools/fuzz_qp_edges.py"""
import sys, warnings, numpy as np
sys.path.insert(0, ".")
from mpc_amd import solvers
from mpc_amd.solvers import matrix
from oracle.qp_oracle import solve_qp
solvers.options["show_progress"] = False
rng = np.random.default_rng(3)
nfail = 0
def run(tag, P, q, G=None, h=None, A=None, b=None, expect="optimal"):
    global nfail
    m = lambda a: None if a is None else matrix(np.asarray(a, float), tc='d')
    try:
        r = solvers.qp(m(P), m(np.asarray(q, float).reshape(-1, 1)), m(G), None if h is None else m(np.asarray(h, float).reshape(-1, 1)),
                       m(A), None if b is None else m(np.asarray(b, float).reshape(-1, 1)))
    except Exception as e:      # noqa: BLE001
        print(f"{tag}: EXCEPTION {type(e).__name__}: {e}"); nfail += expect != "raises"; return
    st = r["status"]; x = np.array(r["x"]).reshape(-1)
    msg = f"{tag}: status {st}"
    ok = st == expect
    if expect == "optimal" and ok:
        s = solve_qp(np.asarray(P, float), np.asarray(q, float).reshape(-1), None if G is None else np.asarray(G, float),
                     None if h is None else np.asarray(h, float).reshape(-1), None if A is None else np.asarray(A, float),
                     None if b is None else np.asarray(b, float).reshape(-1))
        e = np.abs(x - s["x"]).max(); eo = abs(r["primal objective"] - s["primal objective"])
        ok = s["status"] == "optimal" and e <= 1e-6 * max(1, np.abs(s["x"]).max()) and eo <= 1e-8 * max(abs(s["primal objective"]), 1e-9)
        msg += f" |x - oracle| {e:.1e} obj err {eo:.1e}"
    print(msg + ("" if ok else "   <-- UNEXPECTED")); nfail += int(not ok)
n = 6
M = rng.standard_normal((n, n)); P = M @ M.T + 0.1 * np.eye(n); q = rng.standard_normal(n)
run("unconstrained", P, q)
run("one variable, one row", [[2.0]], [1.0], [[1.0]], [-3.0])
run("one variable, unconstrained", [[2.0]], [1.0])
A = rng.standard_normal((2, n)); b = rng.standard_normal(2)
run("equalities only", P, q, None, None, A, b)
G = rng.standard_normal((9, n)); h = np.abs(rng.standard_normal(9)) + 0.1
run("inequalities only", P, q, G, h)
run("duplicate rows", P, q, np.vstack([G, G]), np.concatenate([h, h]))
run("duplicate equalities (consistent)", P, q, None, None, np.vstack([A, A]), np.concatenate([b, b]))
run("infeasible rows", P, q, [[1.0] + [0.0] * (n - 1), [-1.0] + [0.0] * (n - 1)], [-1.0, -1.0], expect="unknown")
Psd = np.zeros((n, n)); Psd[0, 0] = 1.0
run("semidefinite, bounded by a box", Psd, q, np.vstack([np.eye(n), -np.eye(n)]), np.ones(2 * n))
run("semidefinite, unbounded below", Psd, q, None, None, expect="unknown")
run("zero Hessian LP in a box", np.zeros((n, n)), q, np.vstack([np.eye(n), -np.eye(n)]), np.ones(2 * n))
run("huge scales", 1e8 * P, 1e8 * q, 1e-4 * G, 1e-4 * h)
run("tiny scales", 1e-8 * P, 1e-8 * q, G, h)
print("FAILURES", nfail)
