"""GPU parity tests of the general-constraint MPC path (BASELINE.json config 5 at reduced horizon): input box +
state box + polytopic rows, one shared Hessian and constraint matrix, per-instance q and h, against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _oracle_rows(prob):
    """Restatement of the condensed rows: G U <= h0 - E x0 (same order as the product, infinite bounds dropped)."""
    from oracle import mpc_oracle as mo
    N, nx, nu = prob["N"], prob["nx"], prob["nu"]
    n = N * nu
    Gam, Phi = mo.rollout_matrices(prob["A"], prob["B"], N)
    I = np.eye(n); Z = np.zeros((n, nx))
    rows = [(I, Z, np.tile(prob["u_max"], N)), (-I, Z, -np.tile(prob["u_min"], N)),
            (Gam, Phi, np.tile(prob["x_max"], N)), (-Gam, -Phi, -np.tile(prob["x_min"], N))]
    Cb = np.kron(np.eye(N), prob["C"])
    rows += [(Cb @ Gam, Cb @ Phi, np.tile(prob["c_max"], N)), (-Cb @ Gam, -Cb @ Phi, -np.tile(prob["c_min"], N))]
    G = np.vstack([r[0] for r in rows]); E = np.vstack([r[1] for r in rows]); h0 = np.concatenate([r[2] for r in rows])
    keep = np.isfinite(h0)
    return G[keep], E[keep], h0[keep]


@pytest.mark.parametrize("path", ["rows", "dense"])
@pytest.mark.parametrize("N,nx,nu,batch", [(8, 4, 2, 16), (20, 10, 3, 12), (5, 3, 3, 9)])
def test_state_and_polytope_rows_match_oracle(N, nx, nu, batch, path):
    from mpc_amd.configs import long_horizon
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_qp
    prob = long_horizon(batch=batch, seed=3, N=N, nx=nx, nu=nu)
    prob["x0"] *= 1.6 if N >= 8 else 2.4  # far enough out that state and polytope rows bind (some instances infeasible)
    prob["x_max"][0] = np.inf            # a dropped (null) row block
    pl = Planner(nx, nu, N).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.general_path = path
    pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"],
                       C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])
    U = pl.solve(prob["x0"], want_multipliers=True).cpu().numpy().reshape(batch, -1)
    status = pl.info.status.cpu().numpy()
    obj = pl.info.obj.cpu().numpy()
    H, F, f0 = mo.condense(prob)
    G, E, h0 = _oracle_rows(prob)
    assert pl.info.lam.shape[1] == G.shape[0]
    assert pl.general_info() == {"m": G.shape[0], "kernel": path}
    # the rows the library built and pruned on the device are the restatement's, in the same order
    assert np.abs(pl.get("G").cpu().numpy() - G).max() <= 1e-12 * max(1.0, np.abs(G).max())
    assert np.abs(pl.get("E").cpu().numpy() - E).max() <= 1e-12 * max(1.0, np.abs(E).max())
    assert np.array_equal(pl.get("h0").cpu().numpy(), h0)
    nbind = 0
    for i in range(batch):
        q = F @ prob["x0"][i] + f0; h = h0 - E @ prob["x0"][i]
        s = solve_qp(H, q, G, h)
        if s["status"] != "optimal":
            assert status[i] != 0, i
            continue
        assert status[i] == 0, i
        assert _rel(U[i], s["x"]) <= 1e-6, i
        assert abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12), i
        assert (G @ U[i] - h <= 1e-9 * max(1.0, np.abs(h).max())).all()
        nbind += int((np.abs(G @ s["x"] - h) < 1e-8)[2 * N * nu:].sum())
    assert nbind > 0 or N < 8            # state / polytope rows do bind somewhere in the batch (odd-size case: n = 15)


def test_full_horizon_config5_kkt_certificate():
    """BASELINE.json configs[4] at its full size (nx=20, nu=6, N=100: n=600, 13,200 one-sided rows): too large for the
    oracle to finish in seconds, so parity is the solver-independent KKT certificate on every instance."""
    from mpc_amd.configs import long_horizon
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    batch = 8
    prob = long_horizon(batch=batch, seed=1)
    prob["x0"] *= 1.3
    pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"],
                       C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])
    U = pl.solve(prob["x0"], want_multipliers=True).cpu().numpy().reshape(batch, -1)
    st = pl.info.status.cpu().numpy(); z = pl.info.lam.cpu().numpy(); obj = pl.info.obj.cpu().numpy()
    assert pl.general_info()["kernel"] == "rows"         # the default: the constraint-space kernel
    # the general dense kernel on the same data lands on the same vertex
    pl.general_path = "dense"
    Ud = pl.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    std = pl.info.status.cpu().numpy(); objd = pl.info.obj.cpu().numpy()
    assert (st == std).all()
    ok = st == 0
    assert _rel(U[ok], Ud[ok]) <= 1e-8
    assert np.abs(obj[ok] - objd[ok]).max() <= 1e-9 * np.abs(objd[ok]).max()
    H, F, f0 = mo.condense(prob)
    G, E, h0 = _oracle_rows(prob)
    assert z.shape[1] == G.shape[0] == 13200
    assert (st == 0).sum() >= batch // 2
    for i in np.nonzero(st == 0)[0]:
        q = F @ prob["x0"][i] + f0; h = h0 - E @ prob["x0"][i]
        scale = max(1.0, np.abs(q).max())
        assert np.abs(H @ U[i] + q + G.T @ z[i]).max() <= 1e-8 * scale          # stationarity
        assert (G @ U[i] - h).max() <= 1e-9 * max(1.0, np.abs(h).max())          # primal feasibility
        assert z[i].min() >= 0.0                                                  # dual feasibility
        assert abs(z[i] @ (G @ U[i] - h)) <= 1e-8 * scale                         # complementarity
        assert abs(obj[i] - (0.5 * U[i] @ H @ U[i] + q @ U[i])) <= 1e-8 * max(1.0, abs(obj[i]))
        assert (z[i] > 0).sum() > 0


def test_rows_path_second_pass_beyond_96_active_rows():
    """More than 96 rows bind (saturated inputs over most of a 40-step horizon): the constraint-space kernel hands the
    instance to its second pass (capacity 176) and still lands where the general dense kernel does."""
    from mpc_amd.configs import long_horizon
    from mpc_amd.planner import Planner
    batch, N, nx, nu = 6, 40, 10, 4
    prob = long_horizon(batch=batch, seed=2, N=N, nx=nx, nu=nu)
    prob["u_min"] = prob["u_min"] * 0.02; prob["u_max"] = prob["u_max"] * 0.02
    prob["x_min"] = prob["x_min"] * 50; prob["x_max"] = prob["x_max"] * 50
    prob["c_min"] = prob["c_min"] * 50; prob["c_max"] = prob["c_max"] * 50
    out = {}
    for path in ("rows", "dense"):
        pl = Planner(nx, nu, N).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.general_path = path
        pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"],
                           C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])
        U = pl.solve(prob["x0"], want_multipliers=True).cpu().numpy().reshape(batch, -1)
        out[path] = (U, pl.info.status.cpu().numpy(), pl.info.obj.cpu().numpy(), (pl.info.lam > 0).sum(1).cpu().numpy())
    U, st, obj, nact = out["rows"]; Ud, std, objd, nactd = out["dense"]
    assert (st == 0).all() and (std == 0).all()
    assert nact.max() > 96 and (nact == nactd).all()
    assert _rel(U, Ud) <= 1e-8
    assert np.abs(obj - objd).max() <= 1e-9 * np.abs(objd).max()


def test_state_rows_with_one_model_per_instance_match_oracle():
    """State-box and polytopic rows when every instance has its OWN time-varying model (the reference's shape: state rows
    LaunchPlannerQPFormulation.py:109-118 inside a re-linearising loop, CollocationBasedPlanner2D.py:85-365): each instance
    is a dense QP with its own Hessian and rows, built and solved behind mpcqp_factor / mpcqp_solve."""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_qp
    rng = np.random.default_rng(11)
    nx, nu, N, nb, nc = 3, 2, 6, 7, 2
    A = 0.9 * np.eye(nx) + 0.15 * rng.standard_normal((nb, N, nx, nx))
    Bm = rng.standard_normal((nb, N, nx, nu)) * 0.6
    Q = np.diag([2.0, 1.0, 0.5]); R = 0.2 * np.eye(nu); P = 3.0 * Q
    u_min = np.array([-0.8, -0.6]); u_max = np.array([0.7, 0.9])
    x_min = np.array([-0.9, -np.inf, -1.0]); x_max = np.array([0.8, 1.1, np.inf])       # open sides are pruned
    Cm = rng.standard_normal((nc, nx)); c_min = -1.2 * np.ones(nc); c_max = 1.0 * np.ones(nc)
    x0 = rng.uniform(-1.2, 1.2, (nb, nx))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P)
    pl.set_constraints(u_min, u_max, x_min=x_min, x_max=x_max, C=Cm, c_min=c_min, c_max=c_max)
    U = pl.solve(x0, want_multipliers=True).cpu().numpy().reshape(nb, -1)
    st = pl.info.status.cpu().numpy(); obj = pl.info.obj.cpu().numpy()
    assert pl.general_info()["kernel"] == "dense"
    n = N * nu
    nopt = 0
    for i in range(nb):
        prob = dict(N=N, nx=nx, nu=nu, A=A[i], B=Bm[i], Q=Q, R=R, P=P, u_min=u_min, u_max=u_max, C=Cm,
                    x_min=x_min, x_max=x_max, c_min=c_min, c_max=c_max)
        H, F, f0 = mo.condense(prob)
        G, E, h0 = _oracle_rows(prob)
        assert pl.general_info()["m"] == G.shape[0]
        s = solve_qp(H, F @ x0[i] + f0, G, h0 - E @ x0[i])
        if s["status"] != "optimal":
            assert st[i] != 0, i
            continue
        nopt += 1
        assert st[i] == 0, i
        assert _rel(U[i], s["x"]) <= 1e-6, i
        assert abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12), i
    assert nopt >= 3


def test_per_instance_bounds_that_differ_in_finiteness_are_all_kept():
    """One model per instance with per-instance input bounds: a side that is open (+-inf) in model 0 but bounded in another
    model must still constrain that model -- the rows to keep are decided over ALL models, not from model 0's pattern."""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_qp
    rng = np.random.default_rng(23)
    nx, nu, N, nb = 3, 2, 5, 6
    A = 0.9 * np.eye(nx) + 0.1 * rng.standard_normal((nb, N, nx, nx))
    Bm = rng.standard_normal((nb, N, nx, nu)) * 0.6
    Q = np.eye(nx); R = 0.1 * np.eye(nu); P = 2.0 * Q
    u_min = np.tile(np.array([-0.3, -0.4]), (nb, N, 1)); u_max = np.tile(np.array([0.25, 0.35]), (nb, N, 1))
    u_max[0, :, 0] = np.inf; u_min[0, :, 1] = -np.inf          # open in model 0 ...
    u_max[3, :, 1] = np.inf                                     # ... and another pattern in model 3
    x_min = np.array([-2.0, -np.inf, -2.0]); x_max = np.array([2.0, 2.0, np.inf])
    x0 = rng.uniform(-1.5, 1.5, (nb, nx))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P)
    pl.set_constraints(u_min, u_max, x_min=x_min, x_max=x_max)
    U = pl.solve(x0).cpu().numpy().reshape(nb, -1)
    st = pl.info.status.cpu().numpy()
    assert pl.general_info()["m"] == 2 * N * nu + 4 * N          # every input row (bounded in some model) + the 4 finite state sides
    binds = 0
    for i in range(nb):
        prob = dict(N=N, nx=nx, nu=nu, A=A[i], B=Bm[i], Q=Q, R=R, P=P)
        H, F, f0 = mo.condense(prob)
        Gam, Phi = mo.rollout_matrices(A[i], Bm[i], N)
        I = np.eye(N * nu); Z = np.zeros((N * nu, nx))
        G = np.vstack([I, -I, Gam, -Gam]); E = np.vstack([Z, Z, Phi, -Phi])
        h0 = np.concatenate([u_max[i].reshape(-1), -u_min[i].reshape(-1), np.tile(x_max, N), -np.tile(x_min, N)])
        keep = np.isfinite(h0)                                  # THIS model's finite rows: the rest constrain nothing
        G, E, h0 = G[keep], E[keep], h0[keep]
        s = solve_qp(H, F @ x0[i] + f0, G, h0 - E @ x0[i])
        assert s["status"] == "optimal" and st[i] == 0, i
        assert _rel(U[i], s["x"]) <= 1e-6, i
        Ui = U[i].reshape(N, nu)
        assert (Ui <= u_max[i] + 1e-9).all() and (Ui >= u_min[i] - 1e-9).all(), i
        if i > 0:
            binds += int((np.abs(Ui[:, 0] - u_max[i][:, 0]) < 1e-9).sum())
    assert binds > 0            # the bound that model 0 does not have does bind in the others


@pytest.mark.parametrize("n,m,p", [(96, 40, 0), (131, 57, 3), (257, 300, 0)])
def test_blocked_shared_setup_matches_numpy(n, m, p):
    """The multi-workgroup set-up (blocked_setup.hip: scaling kernels, blocked Cholesky and inverse on the MFMA GEMM) leaves
    in the workspace of QP 0 what the one-workgroup phase of dense_qp_kernel leaves: Ps, (Ps + sigma I)^-1, the scaled
    rows, the scalings, sigma and the status.  Sizes off the 32-wide panel on purpose."""
    import ctypes as C
    import torch
    from mpc_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 20))
    P = M @ M.T / n + np.diag(rng.uniform(0.1, 5.0, n))
    G = rng.standard_normal((m, n)) * rng.uniform(0.1, 10.0, (m, 1))
    A = rng.standard_normal((p, n)) if p else None
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    ni = C.c_long(0)
    nd = lib.mpcqp_dense_workspace(n, m, p, C.byref(ni))
    ws = torch.full((nd,), float("nan"), dtype=torch.float64, device=dev)
    iws = torch.zeros(ni.value, dtype=torch.int32, device=dev)
    Pt, Gt = t(P), t(G)
    At = t(A) if p else None
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)
    rc = lib.mpcqp_dense_shared_setup(0, n, m, p, ptr(Pt), ptr(Gt), ptr(At), ptr(ws), ptr(iws),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    w = ws.cpu().numpy()
    nn, mc = n * n, m + p
    V = max(n, mc)
    Ps = w[:nn].reshape(n, n); Hi = w[nn:2 * nn].reshape(n, n)
    sigma, status = w[2 * nn], w[2 * nn + 1]
    Cs = w[4 * nn:4 * nn + mc * n].reshape(mc, n)
    vec = w[4 * nn + mc * n + 2 * nn:]
    d, rn = vec[:n], vec[3 * V:3 * V + mc]
    d_ref = 1.0 / np.sqrt(np.diag(P))
    assert status == 0.0 and abs(sigma - 1e-6) <= 1e-18
    assert np.abs(d - d_ref).max() <= 1e-15 * d_ref.max()
    Ps_ref = P * np.outer(d_ref, d_ref)
    assert np.abs(Ps - Ps_ref).max() <= 1e-15
    rows = np.vstack([A, G]) if p else G
    rn_ref = np.linalg.norm(rows * d_ref, axis=1)
    assert np.abs(rn - rn_ref).max() <= 1e-13 * rn_ref.max()
    assert np.abs(Cs - rows * d_ref / rn_ref[:, None]).max() <= 1e-14
    K = Ps_ref + sigma * np.eye(n)
    assert np.abs(Hi @ K - np.eye(n)).max() <= 1e-10 * np.linalg.cond(K) ** 0.5
    assert np.abs(Hi - Hi.T).max() <= 1e-12 * np.abs(Hi).max()
    # an indefinite Hessian is reported, not factored into NaNs that reach the caller as a solution
    P2 = P.copy(); k = n // 2 + 3                      # a 2 x 2 principal minor with a negative determinant
    P2[k, k + 1] = P2[k + 1, k] = 10.0 * np.sqrt(P[k, k] * P[k + 1, k + 1])
    rc = lib.mpcqp_dense_shared_setup(0, n, m, p, ptr(t(P2)), ptr(Gt), ptr(At), ptr(ws), ptr(iws),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.linalg.eigvalsh(P2).min() < 0 and ws[2 * nn + 1].item() == 3.0


@pytest.mark.parametrize("cond", [1e2, 1e9])
def test_rows_kernel_tail_refines_when_the_inverse_is_not_accurate_enough(cond):
    """The constraint-space kernel certifies optimality of (x, lam) from |I - Ps Hi|_inf (computed by the set-up) and skips
    its refinement when that bound is at rounding level (cond 1e2).  With an ill-conditioned Hessian (cond 1e9 AFTER the
    Jacobi scaling) the bound fails and the refinement round with Ps / Cs themselves has to deliver the answer: both cases
    through the C-ABI (mpcqp_rows_setup / mpcqp_rows_solve) against the oracle."""
    import ctypes as C
    import torch
    from mpc_amd import _lib
    from oracle.qp_oracle import solve_qp
    lib = _lib.load()
    rng = np.random.default_rng(17)
    n, m, B = 48, 70, 6
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = Qo @ np.diag(np.logspace(0, np.log10(cond), n)) @ Qo.T
    P = 0.5 * (P + P.T)
    G = rng.standard_normal((m, n))
    q = rng.standard_normal((B, n)) * 3.0
    h = rng.uniform(0.05, 0.5, (B, m))
    dev = torch.device("cuda:0")
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=dev).contiguous()
    per = C.c_long(0)
    nsh = lib.mpcqp_rows_workspace(n, m, C.byref(per))
    sh = torch.zeros(nsh, dtype=torch.float64, device=dev)
    iw = torch.zeros(n, dtype=torch.int32, device=dev)
    inst = torch.zeros(B * per.value, dtype=torch.float64, device=dev)
    Pt, Gt, qt, ht = t(P), t(G), t(q), t(h)
    x = torch.zeros((B, n), dtype=torch.float64, device=dev); z = torch.zeros((B, m), dtype=torch.float64, device=dev)
    obj = torch.zeros(B, dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    ptr = lambda a: C.c_void_p(a.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.mpcqp_rows_setup(0, n, m, ptr(Pt), ptr(Gt), ptr(sh), ptr(iw), stream) == 0
    assert lib.mpcqp_rows_solve(0, B, n, m, ptr(qt), ptr(ht), ptr(x), ptr(z), ptr(obj), ptr(st), ptr(it), ptr(sh), ptr(inst),
                                stream) == 0
    torch.cuda.synchronize()
    eps = sh[2 * n * n + 2].item()                   # |I - Ps Hi|_inf, beside sigma and the set-up status
    assert (eps < 1e-12) == (cond < 1e5), eps        # which branch of the tail this case exercises
    xs, zs, os_, ss = x.cpu().numpy(), z.cpu().numpy(), obj.cpu().numpy(), st.cpu().numpy()
    nact = 0
    for i in range(B):
        s = solve_qp(P, q[i], G, h[i])
        assert s["status"] == "optimal" and ss[i] == 0, i
        assert np.abs(xs[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()), i
        assert abs(os_[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12), i
        g = P @ xs[i] + q[i] + G.T @ zs[i]
        assert np.abs(g).max() <= 1e-8 * max(1.0, np.abs(q[i]).max()), i
        nact += int((zs[i] > 0).sum())
    assert nact >= B


@pytest.mark.parametrize("nx,nu,N", [(72, 2, 6), (100, 3, 5), (65, 1, 20)])
def test_more_than_64_states_take_the_general_route(nx, nu, N):
    """The box kernels stage x0 in 64 lanes; a model with more states is routed to the general-constraint kernels inside
    mpcqp_factor (q = F x0 + f0 by a GEMM of any depth), and the assembly keeps Q, P in memory once the Phi ping-pong alone
    fills the LDS (nx > 71).  Round 2 raised NotImplementedError here.  Input box only, against the oracle."""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    rng = np.random.default_rng(nx)
    A = 0.9 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    prob = dict(nx=nx, nu=nu, N=N, A=A, B=B, Q=np.eye(nx), R=0.1 * np.eye(nu), P=2.0 * np.eye(nx),
                u_min=-0.3 * np.ones(nu), u_max=0.2 * np.ones(nu), x0=rng.uniform(-1, 1, (40, nx)))
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(40, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], N); ub = np.tile(prob["u_max"], N)
    obj = pl.info.obj.cpu().numpy()
    for i in range(0, 40, 4):
        s = solve_box_qp(H, F @ prob["x0"][i] + f0, lb, ub)
        assert np.abs(U[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()), i
        assert abs(obj[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12), i
    assert int(((U >= ub - 1e-9) | (U <= lb + 1e-9)).sum()) > 40          # the bounds do bind


@pytest.mark.parametrize("blocks,used", [([(10, True), (16, True), (7, False)], True),     # even blocks with their negatives + one-sided rows
                                         ([(9, True), (12, True)], False),                # an odd block: negatives do not pair up two by two
                                         ([(14, False)], True)])                           # no negatives at all: every pair is visited
def test_rows_stacked_with_their_negatives_are_swept_once(blocks, used):
    """The reference stacks two-sided bounds as a row and its negative (LaunchPlannerQPFormulation.py:96-118).  The set-up of the
    constraint-space kernel finds such pairs in the dual Hessian (Gram[a][r'] = -Gram[a][r]), and a step's residual sweep then
    reads one column of the two.  Detection (the flag, the number of row pairs a sweep visits) and the answers against the
    oracle, with the shortcut in use, refused (pairs misaligned) and without any mirrored row."""
    import ctypes as C
    import torch
    from mpc_amd import _lib
    from oracle.qp_oracle import solve_qp
    lib = _lib.load()
    rng = np.random.default_rng(23)
    n, B = 30, 5
    parts, hs, mirrored = [], [], 0
    for rows, both in blocks:
        A = rng.standard_normal((rows, n))
        parts.append(A); hs.append(rng.uniform(0.05, 0.4, (B, rows)))
        if both:
            parts.append(-A); hs.append(rng.uniform(0.05, 0.4, (B, rows))); mirrored += rows
    G = np.vstack(parts); h = np.hstack(hs)
    m = G.shape[0]
    if m & 1:
        G = np.vstack([G, rng.standard_normal((1, n))]); h = np.hstack([h, rng.uniform(0.05, 0.4, (B, 1))]); m += 1
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = Qo @ np.diag(np.logspace(0, 2, n)) @ Qo.T
    P = 0.5 * (P + P.T)
    q = rng.standard_normal((B, n)) * 4.0
    dev = torch.device("cuda:0")
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=dev).contiguous()
    per = C.c_long(0)
    nsh = lib.mpcqp_rows_workspace(n, m, C.byref(per))
    sh = torch.zeros(nsh, dtype=torch.float64, device=dev)
    iw = torch.zeros(n, dtype=torch.int32, device=dev)
    inst = torch.zeros(B * per.value, dtype=torch.float64, device=dev)
    Pt, Gt, qt, ht = t(P), t(G), t(q), t(h)
    x = torch.zeros((B, n), dtype=torch.float64, device=dev); z = torch.zeros((B, m), dtype=torch.float64, device=dev)
    obj = torch.zeros(B, dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    ptr = lambda a: C.c_void_p(a.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.mpcqp_rows_setup(0, n, m, ptr(Pt), ptr(Gt), ptr(sh), ptr(iw), stream) == 0
    assert lib.mpcqp_rows_solve(0, B, n, m, ptr(qt), ptr(ht), ptr(x), ptr(z), ptr(obj), ptr(st), ptr(it), ptr(sh), ptr(inst),
                                stream) == 0
    torch.cuda.synchronize()
    # the ints behind Gram: mirror of every row (m) | mirror pair of every pair (m/2 + 1) | ok, working pairs, -, - | list
    nd = ((m + 2 * (m // 2 + 1) + 4 + 1) // 2 + 1)
    blk = sh[nsh - nd:].view(torch.int32).cpu().numpy()
    mir, ok, nwork = blk[:m], int(blk[m + m // 2 + 1]), int(blk[m + m // 2 + 2])
    assert int((mir >= 0).sum()) == 2 * mirrored          # every stacked row found its negative, nothing else did
    assert ok == (1 if used else 0)
    assert nwork == (m // 2 - mirrored // 2 if used else m // 2)
    xs, zs, os_, ss = x.cpu().numpy(), z.cpu().numpy(), obj.cpu().numpy(), st.cpu().numpy()
    for i in range(B):
        s = solve_qp(P, q[i], G, h[i])
        assert s["status"] == "optimal" and ss[i] == 0, i
        assert np.abs(xs[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max()), i
        assert abs(os_[i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-12), i
        assert (G @ xs[i] - h[i]).max() <= 1e-9
        assert (zs[i] >= -1e-12).all()
    assert int((zs > 0).sum()) >= B


def test_warm_start_of_the_constraint_space_kernel_lands_on_the_cold_solution():
    """warm_start in the general-constraint mode (mpcqp_rows_solve_mpc_warm): the solve starts from the working set the previous
    solve of the same batch ended with.  (a) the same data again: confirmed without a step; (b) nearby data: the same minimiser
    and multipliers as a cold solve of that data, in fewer steps; (c) data on the other side (x0 mirrored: every kept row
    has the wrong sign): the wrong-signed members are released and the answer is still the cold one; (d) each against the
    oracle; (e) another batch size, or a new factorisation, falls back to a cold solve; the dense kernel refuses."""
    import torch
    from mpc_amd.configs import long_horizon
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_qp
    N, nx, nu, batch = 20, 10, 3, 24
    prob = long_horizon(batch=batch, seed=11, N=N, nx=nx, nu=nu)
    prob["x0"] *= 1.5

    def planner(path="rows"):
        pl = Planner(nx, nu, N).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
        pl.general_path = path
        return pl.set_constraints(prob["u_min"], prob["u_max"], x_min=prob["x_min"], x_max=prob["x_max"],
                                  C=prob["C"], c_min=prob["c_min"], c_max=prob["c_max"])

    rng = np.random.default_rng(5)
    xa = prob["x0"]
    xb = xa + 0.03 * rng.standard_normal(xa.shape) * np.abs(xa).max()
    xc = -xa
    cold = planner()
    ref = {}
    for name, x in (("a", xa), ("b", xb), ("c", xc)):
        U = cold.solve(x, want_multipliers=True).cpu().numpy().reshape(batch, -1)
        ref[name] = (U, cold.info.lam.cpu().numpy(), cold.info.status.cpu().numpy(), cold.info.obj.cpu().numpy(), cold.info.polish_steps.cpu().numpy())
    assert (ref["a"][4] > 0).sum() >= batch // 2                   # rows do bind
    # solved means feasible: an LP (largest common slack of all rows) says which instances have a feasible point at all --
    # one of these is infeasible by 0.03 and used to come back SOLVED with multipliers of 1e15 (the last row passed the
    # dependency test by a hair); now the kernel checks that what it calls solved holds its own active rows
    from scipy.optimize import linprog
    H, F, f0 = mo.condense(prob)
    G, E, h0 = _oracle_rows(prob)
    nvar = G.shape[1]
    cost = np.zeros(nvar + 1); cost[-1] = -1.0
    Aub = np.hstack([G, np.ones((G.shape[0], 1))])
    for name, x in (("a", xa), ("b", xb), ("c", xc)):
        for i in range(batch):
            r = linprog(cost, A_ub=Aub, b_ub=h0 - E @ x[i], bounds=[(None, None)] * nvar + [(None, 10.0)], method="highs")
            if r.x[-1] < -1e-6:
                assert ref[name][2][i] != 0, (name, i, r.x[-1])
            elif r.x[-1] > 1e-6:
                assert ref[name][2][i] == 0, (name, i, r.x[-1])
    assert (ref["a"][2] != 0).sum() >= 3
    pl = planner()
    pl.solve(xa)
    seen = {}
    for name, x in (("a", xa), ("b", xb), ("c", xc), ("b", xb)):
        U = pl.solve(x, warm_start=True, want_multipliers=True).cpu().numpy().reshape(batch, -1)
        Ur, lr, sr, objr, itr = ref[name]
        st = pl.info.status.cpu().numpy(); it = pl.info.polish_steps.cpu().numpy()
        assert np.array_equal(st, sr), name
        ok = sr == 0
        assert _rel(U[ok], Ur[ok]) <= 1e-9 and _rel(pl.info.lam.cpu().numpy()[ok], lr[ok]) <= 1e-8, name
        assert _rel(pl.info.obj.cpu().numpy()[ok], objr[ok]) <= 1e-10, name
        seen.setdefault(name, []).append((it[ok].sum(), itr[ok].sum()))
    assert seen["a"][0][0] == 0                                     # the same data: nothing to do
    assert seen["b"][0][0] < 0.5 * seen["b"][0][1]                  # nearby data: a fraction of the cold steps
    # against the oracle (the cold answers are, elsewhere; here the warm ones of the last solve)
    for i in range(0, batch, 5):
        s = solve_qp(H, F @ xb[i] + f0, G, h0 - E @ xb[i])
        if s["status"] == "optimal":
            assert _rel(U[i], np.asarray(s["x"]).reshape(-1)) <= 1e-6, i
    # another batch size: no kept sets for it -> a cold solve, same answer
    U8 = pl.solve(xb[:8], warm_start=True).cpu().numpy().reshape(8, -1)
    assert np.array_equal(pl.info.polish_steps.cpu().numpy(), ref["b"][4][:8])
    assert _rel(U8[ref["b"][2][:8] == 0], ref["b"][0][:8][ref["b"][2][:8] == 0]) <= 1e-9
    with pytest.raises(ValueError, match="dense kernel keeps no state"):
        planner("dense").solve(xa, warm_start=True)
