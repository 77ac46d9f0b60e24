"""GPU parity tests of the one-model-per-instance path (BASELINE.json config 4): batched assembly, batched
factorisation and the dual active-set solve, against the CPU oracle on the same seeded inputs; then the
closed-loop LTV driver (kinematic bicycle, re-linearisation about the shifted plan, warm start)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMAL_RTOL = 1e-6
OBJ_RTOL = 1e-8


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _oracle_instance(A, B, Q, R, P, x_ref, u_ref, lb, ub, x0):
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    N, nx, nu = A.shape[0], A.shape[1], B.shape[2]
    H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=P), x_ref=x_ref, u_ref=u_ref)
    s = solve_box_qp(H, F @ x0 + f0, lb.reshape(-1), ub.reshape(-1))
    assert s["kkt"]["max"] <= 1e-9
    return s


def test_per_instance_models_match_oracle():
    import torch
    from mpc_amd.planner import Planner
    rng = np.random.default_rng(11)
    Bn, N, nx, nu = 24, 12, 4, 2
    A0 = np.eye(nx) + 0.1 * np.diag(np.ones(nx - 1), 1)
    B0 = np.array([[0.0, 0.0], [0.005, 0.0], [0.0, 0.02], [0.1, 0.1]])
    A = A0 + 0.05 * rng.standard_normal((Bn, N, nx, nx))
    Bm = B0 + 0.02 * rng.standard_normal((Bn, N, nx, nu))
    Q = np.diag([1.0, 2.0, 0.5, 0.1]); R = np.diag([0.05, 0.2]); P = 3.0 * Q
    x_ref = 0.3 * rng.standard_normal((Bn, N, nx)); u_ref = 0.1 * rng.standard_normal((Bn, N, nu))
    lb = -0.4 - 0.2 * rng.random((Bn, N, nu)); ub = 0.4 + 0.2 * rng.random((Bn, N, nu))
    x0 = rng.standard_normal((Bn, nx))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P, x_ref=x_ref, u_ref=u_ref).set_constraints(lb, ub)
    U = pl.solve(x0, want_multipliers=True).cpu().numpy().reshape(Bn, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    obj = pl.info.obj.cpu().numpy(); lam = pl.info.lam.cpu().numpy()
    nact = 0
    for i in range(Bn):
        s = _oracle_instance(A[i], Bm[i], Q, R, P, x_ref[i], u_ref[i], lb[i], ub[i], x0[i])
        assert _rel(U[i], s["x"]) <= PRIMAL_RTOL, i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * max(abs(s["primal objective"]), 1e-300), i
        act = (np.abs(s["x"] - lb[i].reshape(-1)) < 1e-9) | (np.abs(s["x"] - ub[i].reshape(-1)) < 1e-9)
        assert ((lam[i] != 0) == act).all() or _rel(U[i], s["x"]) <= 1e-10
        nact += act.sum()
    assert nact > Bn            # the bounds do bind in this test
    # trajectory recovery with one (time-varying) model per instance
    X = pl.predict(x0, U).cpu().numpy()
    for i in (0, 5, 23):
        xk = x0[i].copy()
        for k in range(N):
            xk = A[i, k] @ xk + Bm[i, k] @ U[i].reshape(N, nu)[k]
            assert np.abs(X[i, k] - xk).max() <= 1e-12 * max(1.0, np.abs(xk).max())
    # per-instance LTI (stage dimension 1) takes the same path
    pl2 = Planner(nx, nu, N).set_model(A[:, :1], Bm[:, :1]).set_cost(Q, R, P).set_constraints(lb[0, 0], ub[0, 0])
    U2 = pl2.solve(x0).cpu().numpy().reshape(Bn, -1)
    for i in (0, 7, 23):
        s = _oracle_instance(np.repeat(A[i, :1], N, 0), np.repeat(Bm[i, :1], N, 0), Q, R, P, None, None,
                             np.tile(lb[0, 0], (N, 1)), np.tile(ub[0, 0], (N, 1)), x0[i])
        assert _rel(U2[i], s["x"]) <= PRIMAL_RTOL, i


def _np_jac(x, u, dt, L):
    psi, v, delta = x[2], x[3], u[1]
    A = np.eye(4)
    A[0, 2] = -dt * v * np.sin(psi); A[0, 3] = dt * np.cos(psi)
    A[1, 2] = dt * v * np.cos(psi); A[1, 3] = dt * np.sin(psi)
    A[2, 3] = dt * np.tan(delta) / L
    B = np.zeros((4, 2)); B[2, 1] = dt * v / (L * np.cos(delta) ** 2); B[3, 0] = dt
    return A, B


def _np_step(x, u, dt, L):
    return np.array([x[0] + dt * x[3] * np.cos(x[2]), x[1] + dt * x[3] * np.sin(x[2]),
                     x[2] + dt * x[3] / L * np.tan(u[1]), x[3] + dt * u[0]])


def test_bicycle_closed_loop_matches_oracle_and_converges():
    import torch
    from mpc_amd.configs import make_config
    from mpc_amd.ltv import Bicycle, ClosedLoopLTV
    cfg = make_config(4, batch=48, seed=5)
    N, dt, L = cfg["N"], cfg["dt"], cfg["L"]
    dev = torch.device("cuda:0")
    loop = ClosedLoopLTV(Bicycle(L, dt), N, cfg["Q"], cfg["R"], cfg["P"], cfg["u_min"], cfg["u_max"], device=dev)
    x = torch.as_tensor(cfg["x0"], device=dev)
    Bn = x.shape[0]
    ks = torch.arange(1, N + 1, device=dev, dtype=torch.float64)
    y0 = x[:, 1].abs().clone()
    steps_used = []
    for it in range(60):
        x_ref = torch.zeros((Bn, N, 4), dtype=torch.float64, device=dev)
        x_ref[:, :, 0] = x[:, :1] + cfg["v_ref"] * dt * ks
        x_ref[:, :, 3] = cfg["v_ref"]
        u0 = loop.step(x, x_ref)
        assert (loop.pl.info.status == 0).all(), it
        steps_used.append(float(loop.pl.info.polish_steps.float().mean()))
        if it < 6:          # the QP of this step against the oracle, from the same nominal trajectory
            last = {k: v.cpu().numpy() for k, v in loop.last.items()}
            obj = loop.pl.info.obj.cpu().numpy()
            for i in (0, 13, 31, 47):
                xb = [last["xbar"][i, 0]]
                for k in range(N):
                    xb.append(_np_step(xb[-1], last["ubar"][i, k], dt, L))
                xb = np.array(xb)
                assert np.abs(xb - last["xbar"][i]).max() <= 1e-12
                AB = [_np_jac(xb[k], last["ubar"][i, k], dt, L) for k in range(N)]
                A = np.array([a for a, _ in AB]); Bm = np.array([b for _, b in AB])
                s = _oracle_instance(A, Bm, cfg["Q"], cfg["R"], cfg["P"], last["x_ref"][i] - xb[1:], -last["ubar"][i],
                                     cfg["u_min"] - last["ubar"][i], cfg["u_max"] - last["ubar"][i], np.zeros(4))
                dU = last["dU"][i].reshape(-1)
                assert _rel(dU, s["x"]) <= PRIMAL_RTOL, (it, i)
                assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * max(abs(s["primal objective"]), 1e-12), (it, i)
        assert (u0 <= torch.as_tensor(cfg["u_max"], device=dev) + 1e-9).all()
        assert (u0 >= torch.as_tensor(cfg["u_min"], device=dev) - 1e-9).all()
        x = loop.model.step(x, u0)
    # lane keeping: every rollout ends close to y = 0, heading 0, v = v_ref
    assert (x[:, 1].abs() < 0.05 + 0.05 * y0).all()
    assert (x[:, 2].abs() < 0.05).all()
    assert ((x[:, 3] - cfg["v_ref"]).abs() < 0.2).all()
    # the shifted multipliers are a useful guess: late steps need (almost) no active-set changes
    assert np.mean(steps_used[-10:]) <= 1.0


def test_full_size_closed_loop_keeps_lane_bounds_and_optimality():
    """configs[3] at its stated size: 8,192 rollouts, 300 closed-loop steps (every rollout re-linearised, assembled, factored
    and solved each step), checked through properties that need no oracle: every step of every rollout reports SOLVED and
    applies an input inside the box (accumulated on the device, one read-back at the end); every rollout ends in its lane;
    the shifted multipliers keep the late steps at (almost) no active-set change; and the LAST step's QP of 64 rollouts
    spread over the batch meets the KKT conditions evaluated from that rollout's own H and f0 (mpcqp_get)."""
    import torch
    from mpc_amd.configs import make_config
    from mpc_amd.ltv import Bicycle, ClosedLoopLTV
    cfg = make_config(4, seed=3)
    Bn = cfg["x0"].shape[0]
    assert Bn == 8192
    N, dt, L = cfg["N"], cfg["dt"], cfg["L"]
    dev = torch.device("cuda:0")
    loop = ClosedLoopLTV(Bicycle(L, dt), N, cfg["Q"], cfg["R"], cfg["P"], cfg["u_min"], cfg["u_max"], device=dev)
    x = torch.as_tensor(cfg["x0"], device=dev)
    y0 = x[:, 1].abs().clone()
    ahead = (cfg["v_ref"] * dt * torch.arange(1, N + 1, device=dev, dtype=torch.float64)).reshape(1, N)
    x_ref = torch.zeros((Bn, N, 4), dtype=torch.float64, device=dev)
    x_ref[:, :, 3] = cfg["v_ref"]
    umin = torch.as_tensor(cfg["u_min"], device=dev); umax = torch.as_tensor(cfg["u_max"], device=dev)
    worst_status = torch.zeros((), dtype=torch.int32, device=dev)
    worst_excess = torch.zeros((), dtype=torch.float64, device=dev)
    late_steps = torch.zeros((), dtype=torch.float64, device=dev)
    for it in range(300):
        torch.add(x[:, :1], ahead, out=x_ref[:, :, 0])
        u0 = loop.step(x, x_ref)
        worst_status = torch.maximum(worst_status, loop.pl.info.status.max())
        worst_excess = torch.maximum(worst_excess, torch.maximum(u0 - umax, umin - u0).max())
        if it >= 290:
            late_steps += loop.pl.info.polish_steps.double().mean()
        x = loop.model.step(x, u0)
    assert int(worst_status) == 0
    assert float(worst_excess) <= 1e-9
    assert float(late_steps) / 10 <= 1.0
    assert bool((x[:, 1].abs() < 0.05 + 0.05 * y0).all()) and bool((x[:, 2].abs() < 0.05).all())
    assert bool(((x[:, 3] - cfg["v_ref"]).abs() < 0.2).all())
    # KKT of the last QP (increments about the plan; x0 = 0, so q = f0), from the rollout's own assembled matrices
    dU = loop.last["dU"].reshape(Bn, -1)
    lb = (umin - loop.last["ubar"]).reshape(Bn, -1); ub = (umax - loop.last["ubar"]).reshape(Bn, -1)
    for i in range(0, Bn, 128):
        H = loop.pl.get("H", model=i); q = loop.pl.get("f0", model=i)
        u = dU[i]; g = H @ u + q
        sc = max(float(q.abs().max()), 1.0)
        assert bool((u <= ub[i] + 1e-9).all()) and bool((u >= lb[i] - 1e-9).all()), i
        free = (u < ub[i] - 1e-9) & (u > lb[i] + 1e-9)
        assert float((g.abs() * free).max()) <= 1e-8 * sc, i
        assert bool(((g <= 1e-8 * sc) | (u < ub[i] - 1e-9)).all()) and bool(((g >= -1e-8 * sc) | (u > lb[i] + 1e-9)).all()), i


def test_closed_loop_graph_replay_matches_eager_steps():
    """One closed-loop step recorded as a HIP graph (linearise -> assemble -> factor -> solve -> apply -> shift, no host
    work in between) drives the rollouts exactly like the eager step() does."""
    import torch
    from mpc_amd.configs import make_config
    from mpc_amd.ltv import Bicycle, ClosedLoopLTV
    cfg = make_config(4, batch=64, seed=7)
    N, dt, L = cfg["N"], cfg["dt"], cfg["L"]
    dev = torch.device("cuda:0")
    ks = torch.arange(1, N + 1, device=dev, dtype=torch.float64)

    def ref_of(x):
        x_ref = torch.zeros((x.shape[0], N, 4), dtype=torch.float64, device=dev)
        x_ref[:, :, 0] = x[:, :1] + cfg["v_ref"] * dt * ks
        x_ref[:, :, 3] = cfg["v_ref"]
        return x_ref

    def run(graph_after):
        loop = ClosedLoopLTV(Bicycle(L, dt), N, cfg["Q"], cfg["R"], cfg["P"], cfg["u_min"], cfg["u_max"], device=dev)
        x = torch.as_tensor(cfg["x0"], device=dev)
        us = []
        for it in range(12):
            x_ref = ref_of(x)
            if graph_after is not None and it == graph_after:
                loop.capture(x, x_ref)
            if graph_after is not None and it >= graph_after:
                u0 = loop.replay(x, x_ref).clone()
            else:
                u0 = loop.step(x, x_ref)
            assert (loop.pl.info.status == 0).all(), it
            us.append(u0.cpu().numpy())
            x = loop.model.step(x, u0)
        return np.array(us), x.cpu().numpy()

    u_eager, x_eager = run(None)
    u_graph, x_graph = run(4)
    assert np.abs(u_eager - u_graph).max() <= 1e-12
    assert np.abs(x_eager - x_graph).max() <= 1e-12


@pytest.mark.parametrize("N,nx,nu", [(4, 3, 2), (16, 4, 4), (19, 5, 3), (30, 3, 3)])
def test_per_instance_factor_tile_counts(N, nx, nu):
    """n = 8, 64, 57 (one, four and four padded 16 x 16 tiles of the wave-per-model factor kernel) and n = 90 (the
    256-thread kernel): scaled Hessian, its explicit inverse and the solve, model by model."""
    from mpc_amd.planner import Planner
    rng = np.random.default_rng(100 + N)
    Bn = 8
    A0 = 0.9 * np.eye(nx) + 0.1 * np.diag(np.ones(nx - 1), 1)
    A = A0 + 0.05 * rng.standard_normal((Bn, N, nx, nx))
    Bm = 0.3 * rng.standard_normal((Bn, N, nx, nu))
    Q = np.diag(0.5 + rng.random(nx)); R = np.diag(0.05 + 0.1 * rng.random(nu)); P = 2.0 * Q
    lb = -0.3 - 0.2 * rng.random((Bn, N, nu)); ub = 0.3 + 0.2 * rng.random((Bn, N, nu))
    x0 = rng.standard_normal((Bn, nx))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P).set_constraints(lb, ub)
    U = pl.solve(x0).cpu().numpy().reshape(Bn, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    obj = pl.info.obj.cpu().numpy()
    n = N * nu
    for i in range(Bn):
        H = pl.get("H", model=i).cpu().numpy().reshape(n, n)
        d = pl.get("d", model=i).cpu().numpy().reshape(n)
        Hs = pl.get("Hs", model=i).cpu().numpy().reshape(n, n)
        Hi = pl.get("Hi", model=i).cpu().numpy().reshape(n, n)
        assert np.abs(Hs - d[:, None] * H * d[None, :]).max() <= 1e-13 * np.abs(Hs).max()
        assert np.abs(Hs @ Hi - np.eye(n)).max() <= 1e-9
        s = _oracle_instance(A[i], Bm[i], Q, R, P, None, None, lb[i], ub[i], x0[i])
        assert _rel(U[i], s["x"]) <= PRIMAL_RTOL, i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * max(abs(s["primal objective"]), 1e-300), i


@pytest.mark.parametrize("nx,nu,N,tv", [(4, 2, 20, True), (3, 3, 7, True), (5, 1, 33, False), (8, 2, 9, True), (1, 1, 5, True)])
def test_structured_assembly_matches_the_prediction_matrix_form(nx, nu, N, tv):
    """One model per instance with nx <= 8 is assembled WITHOUT the prediction matrices (assemble_wave_kernel: the backward
    recursion P_k = Q_k + A_{k+1}' P_{k+1} A_{k+1} and one chain per column of H).  H, F, f0 must be those of
    H = Gamma' Qbar Gamma + Rbar, F = Gamma' Qbar Phi, f0 = -Gamma' Qbar x_ref - Rbar u_ref (the restatement of
    LaunchPlannerQPFormulation.py:49-89,130-162), per model, with references per model; Gamma itself is still there for
    whoever asks for it."""
    import torch
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    rng = np.random.default_rng(100 * nx + N)
    nb = 5
    shape_a = (nb, N if tv else 1, nx, nx)         # (batch, 1, ...): one time-invariant model per instance
    shape_b = (nb, N if tv else 1, nx, nu)
    A = 0.8 * np.eye(nx) + 0.3 * rng.standard_normal(shape_a)
    Bm = rng.standard_normal(shape_b)
    Mq = rng.standard_normal((nx, nx)); Q = Mq @ Mq.T + 0.1 * np.eye(nx)
    Mp = rng.standard_normal((nx, nx)); P = Mp @ Mp.T + 0.5 * np.eye(nx)
    Mr = rng.standard_normal((nu, nu)); R = Mr @ Mr.T + 0.2 * np.eye(nu)
    xr = rng.standard_normal((nb, N, nx)); ur = rng.standard_normal((nb, N, nu))
    pl = Planner(nx, nu, N).set_model(A, Bm).set_cost(Q, R, P, x_ref=xr, u_ref=ur)
    pl.set_constraints(-np.ones(nu), np.ones(nu)).setup()
    Ai = lambda i: A[i] if tv else A[i, 0]
    Bi = lambda i: Bm[i] if tv else Bm[i, 0]
    for i in range(nb):
        prob = dict(N=N, nx=nx, nu=nu, A=Ai(i), B=Bi(i), Q=Q, R=R, P=P)
        H, F, f0 = mo.condense(prob, x_ref=xr[i], u_ref=ur[i])
        Gam, Phi = mo.rollout_matrices(Ai(i), Bi(i), N)
        Hd = pl.get("H", model=i).cpu().numpy()
        assert np.array_equal(Hd, Hd.T)
        assert np.abs(Hd - H).max() <= 1e-12 * np.abs(H).max(), i
        assert np.abs(pl.get("F", model=i).cpu().numpy() - F).max() <= 1e-12 * max(1.0, np.abs(F).max()), i
        assert np.abs(pl.get("f0", model=i).cpu().numpy() - f0).max() <= 1e-12 * max(1.0, np.abs(f0).max()), i
        assert np.abs(pl.get("Phi", model=i).cpu().numpy() - Phi).max() <= 1e-12 * max(1.0, np.abs(Phi).max()), i
    # Gamma on request (the first call runs the prediction-matrix assembly)
    i = nb - 1
    Gam, _ = mo.rollout_matrices(Ai(i), Bi(i), N)
    assert np.abs(pl.get("Gamma", model=i).cpu().numpy() - Gam).max() <= 1e-12 * max(1.0, np.abs(Gam).max())
    H, _, _ = mo.condense(dict(N=N, nx=nx, nu=nu, A=Ai(i), B=Bi(i), Q=Q, R=R, P=P))
    assert np.abs(pl.get("H", model=i).cpu().numpy() - H).max() <= 1e-12 * np.abs(H).max()
