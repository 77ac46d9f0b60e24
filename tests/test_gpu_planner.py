"""GPU parity tests of the MPC hot path through the C-ABI (libmpcqp.so via ctypes).

Bar (BASELINE.json north star): primal within 1e-6 relative, objective within 1e-8
relative of the certified optimum, in fp64."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMAL_RTOL = 1e-6
OBJ_RTOL = 1e-8


def _planner(prob, split=True):
    from mpc_amd.planner import Planner
    pl = Planner(prob["nx"], prob["nu"], prob["N"])
    pl.split = split                # False: the kernels that take the instance as a whole (what a coupled model runs)
    pl.set_model(prob["A"], prob["B"])
    pl.set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"])
    return pl


def _prob_from_golden(g):
    return dict(nx=g["A"].shape[0], nu=g["B"].shape[1], N=int(g["N"]), A=g["A"], B=g["B"], Q=g["Q"], R=g["R"],
                P=g["P"], u_min=g["u_min"], u_max=g["u_max"], x0=g["x0"])


def _rel(a, b):
    return np.abs(a - b).max(axis=-1) / np.maximum(np.abs(b).max(axis=-1), 1e-300)


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_assembly_matches_oracle(cfg, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mpc_cfg{cfg}.npz"))
    prob = _prob_from_golden(g)
    pl = _planner(prob).setup()
    H = pl.get("H").cpu().numpy(); F = pl.get("F").cpu().numpy(); f0 = pl.get("f0").cpu().numpy()
    sc = np.abs(g["H"]).max()
    assert np.abs(H - g["H"]).max() <= 1e-12 * sc
    assert np.array_equal(H, H.T)
    assert np.abs(F - g["F"]).max() <= 1e-12 * np.abs(g["F"]).max()
    assert np.abs(f0 - g["f0"]).max() <= 1e-12
    from oracle.mpc_oracle import rollout_matrices
    Gam, Phi = rollout_matrices(prob["A"], prob["B"], prob["N"])
    assert np.abs(pl.get("Gamma").cpu().numpy() - Gam).max() <= 1e-13 * np.abs(Gam).max()
    assert np.abs(pl.get("Phi").cpu().numpy() - Phi).max() <= 1e-13 * np.abs(Phi).max()


# split: configs[2] (cfg 3) is four independent chains and is solved component by component unless told otherwise; the
# other two are coupled, so the flag changes nothing for them
SPLIT_CASES = [(1, True), (2, True), (3, True), (3, False)]


@pytest.mark.parametrize("admm", [0, 5])          # active-set kernels alone (the default) / after five ADMM iterations
@pytest.mark.parametrize("cfg,split", SPLIT_CASES)
def test_solve_matches_golden(cfg, split, admm, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mpc_cfg{cfg}.npz"))
    prob = _prob_from_golden(g)
    pl = _planner(prob, split)
    U = pl.solve(prob["x0"], want_multipliers=True, max_iter=admm).cpu().numpy().reshape(len(prob["x0"]), -1)
    info = pl.info
    assert (info.iters.cpu().numpy() == admm).all()
    assert (info.status.cpu().numpy() == 0).all()
    assert _rel(U, g["U"]).max() <= PRIMAL_RTOL
    obj = info.obj.cpu().numpy()
    # the device objective is in Jacobi-scaled variables, which leaves its value unchanged
    assert (np.abs(obj - g["objective"]) <= OBJ_RTOL * np.maximum(np.abs(g["objective"]), 1e-300)).all()
    lam = info.lam.cpu().numpy()
    assert ((lam != 0) == (g["lam"] != 0)).all(), "active set differs from the certified optimum"
    assert np.abs(lam - g["lam"]).max() <= 1e-6 * max(1.0, np.abs(g["lam"]).max())


@pytest.mark.parametrize("admm", [0, 2, 5])
@pytest.mark.parametrize("cfg,batch,scale,split", [(1, 64, 1.0, True), (2, 96, 8.0, True), (3, 200, 1.0, True), (3, 72, 3.0, True),
                                                   (3, 200, 1.0, False), (3, 72, 3.0, False)])
def test_solve_matches_oracle_seeded(cfg, batch, scale, split, admm):
    from mpc_amd.configs import make_config
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    prob = make_config(cfg, batch=batch, seed=7)
    prob["x0"] = prob["x0"] * scale
    pl = _planner(prob, split)
    assert pl.setup().split_info()["components"] == (4 if cfg == 3 and split else 0)
    U = pl.solve(prob["x0"], max_iter=admm).cpu().numpy().reshape(batch, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    obj = pl.info.obj.cpu().numpy()
    step = max(1, batch // (24 if admm == 0 else 8))          # the oracle costs ~50 ms per instance at n = 120
    for i in range(0, batch, step):
        s = solve_box_qp(H, F @ prob["x0"][i] + f0, lb, ub)
        assert s["kkt"]["max"] <= 1e-10
        assert _rel(U[i], s["x"]) <= PRIMAL_RTOL, i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * abs(s["primal objective"]), i
    # every instance: feasibility and KKT residual computed on the host from H itself
    X0 = prob["x0"]
    G = U @ H + (X0 @ F.T + f0)           # gradient rows
    assert (U <= ub + 1e-12).all() and (U >= lb - 1e-12).all()
    free = (U < ub - 1e-9) & (U > lb + 1e-9)
    scale_g = np.abs(X0 @ F.T + f0).max(axis=1, keepdims=True)
    assert (np.abs(G) * free <= 1e-8 * scale_g).all()
    assert ((G <= 1e-8 * scale_g) | (U < ub - 1e-9)).all()      # at upper bound gradient must be <= 0
    assert ((G >= -1e-8 * scale_g) | (U > lb + 1e-9)).all()


def test_admm_only_matches_restatement():
    """polish off: the device iterates equal the NumPy restatement of the same ADMM."""
    from mpc_amd.configs import make_config
    from oracle import mpc_oracle as mo
    prob = make_config(3, batch=48, seed=3)
    pl = _planner(prob).setup(rho=1e-3)
    K = 17
    U = pl.solve(prob["x0"], polish=0, max_iter=K, min_iter=10 ** 6).cpu().numpy().reshape(48, -1)
    assert (pl.info.iters.cpu().numpy() == K).all()
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    fac = mo.factor(H, F, f0, lb, ub, 1e-3)
    z, w = mo.admm_box(fac, prob["x0"], K)
    Uo = z * fac["d"][None, :]
    assert np.abs(U - Uo).max() <= 1e-7 * np.abs(Uo).max()
    zd = pl.get("z", batch=48).cpu().numpy(); wd = pl.get("w", batch=48).cpu().numpy()
    assert np.abs(zd - z).max() <= 1e-7 * np.abs(z).max()
    assert np.abs(wd - w).max() <= 1e-7 * max(np.abs(w).max(), 1e-12)


def test_ltv_references_and_per_stage_bounds():
    """Time-varying (A_k, B_k), non-zero x_ref/u_ref, bounds that differ per stage."""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    rng = np.random.default_rng(5)
    nx, nu, N, B = 5, 2, 12, 40
    A = np.eye(nx)[None] + 0.15 * rng.standard_normal((N, nx, nx))
    Bm = rng.standard_normal((N, nx, nu))
    Q = np.diag(rng.uniform(0.5, 2, nx)); R = np.diag(rng.uniform(0.05, 0.2, nu)); P = 3 * Q
    xr = rng.standard_normal((N, nx)); ur = 0.1 * rng.standard_normal((N, nu))
    umin = -rng.uniform(0.2, 1.0, (N, nu)); umax = rng.uniform(0.2, 1.0, (N, nu))
    x0 = rng.standard_normal((B, nx))
    pl = Planner(nx, nu, N)
    pl.set_model(A, Bm).set_cost(Q, R, P, x_ref=xr, u_ref=ur).set_constraints(umin, umax)
    U = pl.solve(x0).cpu().numpy().reshape(B, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    prob = dict(N=N, nx=nx, nu=nu, A=A, B=Bm, Q=Q, R=R, P=P)
    H, F, f0 = mo.condense(prob, xr, ur)
    assert np.abs(pl.get("f0").cpu().numpy() - f0).max() <= 1e-12 * np.abs(f0).max()
    assert np.abs(pl.get("H").cpu().numpy() - H).max() <= 1e-12 * np.abs(H).max()
    obj = pl.info.obj.cpu().numpy()
    for i in range(0, B, 4):
        s = solve_box_qp(H, F @ x0[i] + f0, umin.ravel(), umax.ravel())
        assert _rel(U[i], s["x"]) <= PRIMAL_RTOL
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * abs(s["primal objective"])


def test_shift_and_warm_start():
    """Receding horizon: shift() moves the kept state one stage forward (last stage repeated); a
    warm-started solve returns the same minimiser as a cold one."""
    from mpc_amd.configs import make_config
    prob = make_config(3, batch=128, seed=21)
    pl = _planner(prob)
    x0 = prob["x0"]
    U0 = pl.solve(x0).cpu().numpy()
    n, nu = pl.n, pl.nu
    z_before = pl.get("z", batch=128).cpu().numpy()
    pl.shift()
    z_after = pl.get("z", batch=128).cpu().numpy()
    assert np.array_equal(z_after[:, : n - nu], z_before[:, nu:])
    assert np.array_equal(z_after[:, n - nu:], z_before[:, n - nu:])
    # closed loop: apply u_0, step the model, solve again warm and cold
    x1 = x0 @ prob["A"].T + U0[:, 0, :] @ prob["B"].T
    Uw = pl.solve(x1, warm_start=True).cpu().numpy()
    steps_w = pl.info.polish_steps.cpu().numpy().astype(float).mean()
    assert (pl.info.status.cpu().numpy() == 0).all()
    Uc = pl.solve(x1, warm_start=False).cpu().numpy()
    steps_c = pl.info.polish_steps.cpu().numpy().astype(float).mean()
    assert np.abs(Uw - Uc).max() <= 1e-9 * max(1.0, np.abs(Uc).max())
    assert steps_w <= steps_c + 0.5
    with pytest.raises(ValueError):
        pl.solve(x1[:64], warm_start=True)          # state belongs to a different batch size


def test_error_conventions():
    """Failures surface as ordinary Exception subclasses (SE3Planner.py:107,198 relies on that)."""
    from mpc_amd.planner import Planner
    pl = Planner(2, 1, 5)
    with pytest.raises(RuntimeError):
        pl.solve(np.zeros((1, 2)))                  # nothing set up
    pl.set_model(np.eye(2), np.ones((2, 1)))
    pl.set_cost(-np.eye(2), -np.eye(1))             # not PD
    pl.set_constraints([-1.0], [1.0])
    with pytest.raises(ArithmeticError):
        pl.solve(np.zeros((1, 2)))
    with pytest.raises(ValueError):
        pl.set_cost(np.eye(3), np.eye(1))
    with pytest.raises(ValueError):
        pl.set_constraints([-1.0], [1.0], c_min=[0.0])      # polytope bounds without C


def test_no_sync_factor_failure_is_reported_as_status():
    """mpcqp_factor with no_sync = 1 cannot tell the host that H is indefinite; the shared model's failure flag stays on
    the device and every instance of the following solve reports MPCQP_MAX_ITER instead of a made-up solution."""
    from mpc_amd import _lib
    from mpc_amd.planner import Planner
    for max_iter in (0, 3):
        pl = Planner(2, 1, 5)
        pl.no_sync = True
        pl.set_model(np.eye(2), np.ones((2, 1)))
        pl.set_cost(-np.eye(2), -np.eye(1))             # not PD
        pl.set_constraints([-1.0], [1.0])
        pl.solve(np.ones((70, 2)), rho=0.1, max_iter=max_iter)   # rho given + no_sync: the factor call does not synchronise
        st = pl.info.status.cpu().numpy()
        assert (st == _lib.MAX_ITER).all(), st
        # the same handle with a definite cost afterwards: the flag is cleared by the next factorisation
        pl.set_cost(np.eye(2), np.eye(1))
        pl.solve(np.ones((70, 2)), rho=0.1, max_iter=max_iter)
        assert (pl.info.status.cpu().numpy() == _lib.SOLVED).all()


@pytest.mark.parametrize("split", [True, False])
def test_two_handles_on_two_streams_do_not_disturb_each_other(split):
    """The serving pattern bench.py reports beside `value`: two handles, two streams, solves in flight together.  Every
    handle owns its queues, counters and hand-over records, so the answers are those of the same solves one by one
    (bit for bit: the cascade is deterministic per instance)."""
    import torch
    from mpc_amd.configs import make_config
    probs = [make_config(3, batch=30000, seed=5), make_config(3, batch=26000, seed=6)]
    pls = [_planner(p, split).setup() for p in probs]
    xs = [torch.as_tensor(1.4 * p["x0"], device="cuda:0") for p in probs]       # long queues in every pass
    ref = [pl.solve(x).clone() for pl, x in zip(pls, xs)]
    # reproducible bit for bit from solve to solve, although the cold-start order (atomic cursors) puts an instance next to
    # different neighbours every time: nothing an instance computes may depend on the other instances of its wave
    for pl, x, r in zip(pls, xs, ref):
        assert torch.equal(pl.solve(x), r)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for pl in pls:
        pl.reuse_outputs = True
    outs = [None, None]
    for rep in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs[i] = pls[i].solve(xs[i])
    torch.cuda.synchronize()
    for i in (0, 1):
        assert int((pls[i].info.status != 0).sum()) == 0
        assert torch.equal(outs[i], ref[i])


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("kw", [{}, {"max_iter": 10}], ids=["cold", "admm-first"])
def test_an_instance_does_not_depend_on_its_neighbours(kw, split):
    """The same instances in another order (other neighbours and another place in the wave): bit for bit the same answers.
    (The warm start once listed its guessed rows in the order of the lane's registers, which depends on the place.)"""
    import torch
    from mpc_amd.configs import make_config
    prob = make_config(3, batch=9000, seed=1)
    pl = _planner(prob, split).setup()
    x = torch.as_tensor(1.5 * prob["x0"], device="cuda:0")
    perm = torch.randperm(9000, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    Ua = pl.solve(x, **kw).clone()
    Ub = pl.solve(x[perm].contiguous(), **kw)
    assert int((pl.info.status != 0).sum()) == 0
    assert torch.equal(Ub, Ua[perm])


def test_two_host_threads_with_their_own_handles():
    """Two host threads, each with its own handle and stream, inside the library at the same time (ctypes releases the
    GIL): the library's only process-wide state is the mutex-guarded table of kernel attributes."""
    import threading
    import torch
    from mpc_amd.configs import make_config
    probs = [make_config(3, batch=20000, seed=8), make_config(2, batch=3000, seed=9)]
    pls = [_planner(p).setup() for p in probs]
    xs = [torch.as_tensor(1.3 * p["x0"], device="cuda:0") for p in probs]
    ref = [pl.solve(x).clone() for pl, x in zip(pls, xs)]
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(20):
                    U = pls[i].solve(xs[i])
                st.synchronize()
                if not torch.equal(U, ref[i]) or int((pls[i].info.status != 0).sum()):
                    errs.append(f"thread {i}: result differs from the single-threaded solve")
        except Exception as e:      # noqa: BLE001
            errs.append(f"thread {i}: {type(e).__name__}: {e}")

    th = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs
    assert not any(t.is_alive() for t in th)


def test_reuse_outputs_returns_the_same_buffers_with_fresh_results():
    """Planner.reuse_outputs (serving loops, bench.py): no allocation per call -- same U / info tensors, new contents."""
    import torch
    from mpc_amd.configs import make_config
    from mpc_amd.planner import Planner
    prob = make_config(3, batch=300, seed=21)
    pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"])
    xa = torch.as_tensor(prob["x0"], device="cuda:0"); xb = torch.as_tensor(prob["x0"][::-1].copy(), device="cuda:0")
    Ua = pl.solve(xa).clone(); Ub = pl.solve(xb).clone()          # fresh tensors per call
    pl.reuse_outputs = True
    U1 = pl.solve(xa); p1 = U1.data_ptr(); s1 = pl.info.status.data_ptr()
    assert torch.equal(U1, Ua)
    U2 = pl.solve(xb)
    assert U2.data_ptr() == p1 and pl.info.status.data_ptr() == s1 and torch.equal(U2, Ub)
    assert torch.equal(U2.flip(0), Ua)                             # the reversed batch, instance by instance
    U3 = pl.solve(xa[:64].contiguous())
    assert U3.data_ptr() != p1 and torch.equal(U3, Ua[:64])       # another batch size, another set of buffers


def test_horizon_beyond_128_inputs_goes_to_the_dense_kernel():
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    rng = np.random.default_rng(4)
    A = np.array([[1.0, 0.1], [0.0, 1.0]]); B = np.array([[0.005, 0.0], [0.1, 0.02]])
    big = Planner(2, 2, 70)                         # N*nu = 140 > 128
    big.set_model(A, B).set_cost(np.eye(2), 0.1 * np.eye(2)).set_constraints([-0.3, -1], [0.3, 1])
    x0 = rng.uniform(-3, 3, (6, 2))
    U = big.solve(x0).cpu().numpy().reshape(6, -1)
    assert (big.info.status.cpu().numpy() == 0).all()
    prob = dict(N=70, nx=2, nu=2, A=A, B=B, Q=np.eye(2), R=0.1 * np.eye(2), P=np.eye(2))
    H, F, f0 = mo.condense(prob)
    for i in range(6):
        s = solve_box_qp(H, F @ x0[i] + f0, np.tile([-0.3, -1.0], 70), np.tile([0.3, 1.0], 70))
        assert np.abs(U[i] - s["x"]).max() <= 1e-6 * max(1.0, np.abs(s["x"]).max())


@pytest.mark.parametrize("split", [True, False])
def test_ragged_and_tiny_batches(split):
    from mpc_amd.configs import make_config
    prob = make_config(3, batch=131, seed=2)
    pl = _planner(prob, split)
    Uall = pl.solve(prob["x0"]).cpu().numpy()
    for b in (1, 15, 17, 63, 65):
        Ub = pl.solve(prob["x0"][:b]).cpu().numpy()
        assert np.abs(Ub - Uall[:b]).max() <= 1e-9 * np.abs(Uall).max(), b


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_predict_matches_prediction_matrices(cfg):
    """x_traj = Gamma u* + Phi x0 (LaunchPlannerQPFormulation.py:176-183) without forming Gamma on the device."""
    from mpc_amd.configs import make_config
    from oracle.mpc_oracle import rollout_matrices
    prob = make_config(cfg, batch=37, seed=9)
    pl = _planner(prob)
    U = pl.solve(prob["x0"])
    X = pl.predict(prob["x0"], U).cpu().numpy().reshape(37, -1)
    Gam, Phi = rollout_matrices(prob["A"], prob["B"], prob["N"])
    Xo = U.cpu().numpy().reshape(37, -1) @ Gam.T + prob["x0"] @ Phi.T
    assert np.abs(X - Xo).max() <= 1e-12 * max(1.0, np.abs(Xo).max())


def _assert_kkt_everywhere(pl, x0, U, u_min, u_max, N):
    """Solver-independent optimality certificate for ALL instances, evaluated on the device from H and F themselves:
    feasibility, stationarity on the free rows, multiplier signs on the active ones, the reported objective.
    Returns the mask of free rows."""
    import torch
    batch = U.shape[0]
    assert bool((pl.info.status == 0).all())
    H, F, f0 = pl.get("H"), pl.get("F"), pl.get("f0")
    q = x0 @ F.t() + f0
    G = U @ H + q
    lb = torch.as_tensor(np.tile(u_min, N), device="cuda:0")
    ub = torch.as_tensor(np.tile(u_max, N), device="cuda:0")
    assert bool((U <= ub + 1e-12).all()) and bool((U >= lb - 1e-12).all())
    scale = q.abs().max(dim=1, keepdim=True).values.clamp_min(1.0)
    free = (U < ub - 1e-9) & (U > lb + 1e-9)
    assert bool(((G.abs() * free) <= 1e-8 * scale).all())
    assert bool(((G <= 1e-8 * scale) | (U < ub - 1e-9)).all())
    assert bool(((G >= -1e-8 * scale) | (U > lb + 1e-9)).all())
    obj = 0.5 * (U * (U @ H)).sum(1) + (q * U).sum(1)
    assert bool(((pl.info.obj - obj).abs() <= 1e-8 * obj.abs().clamp_min(1e-9)).all())
    assert free.shape[0] == batch
    return free


@pytest.mark.parametrize("cfg,batch,x0_scale,split", [(2, 4096, 1.0, True), (3, 65536, 1.0, True), (3, 8192, 2.5, True),
                                                      (3, 16384, 2.5, True), (3, 65536, 1.0, False), (3, 8192, 2.5, False),
                                                      (3, 16384, 2.5, False)])
def test_full_size_batches_satisfy_kkt_everywhere(cfg, batch, x0_scale, split):
    """BASELINE.json's full sizes (4,096 cart-poles; 65,536 quadrotors): the oracle cannot visit every instance in
    seconds, so parity at this size is the solver-independent optimality certificate, evaluated for ALL instances."""
    import torch
    from mpc_amd.configs import make_config
    prob = make_config(cfg, batch=batch, seed=1)
    # scale 2.5: far-out initial states -- large active sets, so most instances outgrow the first pass, many the second and
    # a few the third; at 16,384 instances the second pass's queue is long enough for its 16-lane instantiation
    pl = _planner(prob, split)
    x0 = torch.as_tensor(prob["x0"] * x0_scale, device="cuda:0")
    U = pl.solve(x0).reshape(batch, -1)
    free = _assert_kkt_everywhere(pl, x0, U, prob["u_min"], prob["u_max"], prob["N"])
    # the bounds do bind: on average more than once per instance for the quadrotors, now and then for the cart-poles
    assert int((~free).sum()) > (batch if cfg == 3 else 0)
    if x0_scale > 1.0:
        nact = (~free).sum(1)
        assert int((nact > 16).sum()) > batch // 20 and int(nact.max()) > 32     # every pass of the chain is exercised
        if not split:
            queued = int(pl.get("queues")[0].item())
            assert (queued >= 8192) == (batch >= 16384)      # which second-pass instantiation ran (polish_quad.hip, LONG_QUEUE)
        else:       # per component: two of the four chains carry the load, and some of them bind on more than 16 rows
            rm = torch.as_tensor(pl.split_info()["row_map"], device="cuda:0")
            per = torch.stack([((~free) & (rm == c)).sum(1) for c in range(4)], 1)
            assert int((per > 8).sum()) > batch // 4 and int(per.max()) > 16


def test_long_queue_second_pass_on_an_odd_shape():
    """The 16-lane second pass (queues of 8,192 instances or more) on a shape with odd n and n not a multiple of 16: tight
    input bounds make most instances outgrow the first pass's 16 rows."""
    import torch
    from mpc_amd.planner import Planner
    rng = np.random.default_rng(5)
    nx, nu, N, batch = 5, 3, 13, 12288          # n = 39
    A = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.eye(nx); R = 0.05 * np.eye(nu)
    umin, umax = -0.05 * np.ones(nu), 0.04 * np.ones(nu)
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 2 * Q).set_constraints(umin, umax)
    x0 = torch.as_tensor(rng.uniform(-3, 3, (batch, nx)), device="cuda:0")
    U = pl.solve(x0).reshape(batch, -1)
    free = _assert_kkt_everywhere(pl, x0, U, umin, umax, N)
    assert int(pl.get("queues")[0].item()) >= 8192
    assert int((~free).sum(1).max()) > 16


@pytest.mark.parametrize("nx,nu,N", [(3, 1, 25), (5, 3, 11), (33, 2, 12), (2, 1, 128)])
def test_odd_shapes_match_oracle(nx, nu, N):
    """Shapes off the beaten path of the polish fast path: odd n (no 16-byte row pairs), n not a multiple of 16,
    nx + 1 > 32 (the fast path hands over to the wave-per-instance kernel), n = 128 (largest ADMM tile count)."""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    rng = np.random.default_rng(nx * 100 + N)
    A = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.eye(nx); R = 0.05 * np.eye(nu)
    batch = 70
    x0 = rng.uniform(-2, 2, (batch, nx))
    umin, umax = -0.3 * np.ones(nu), 0.4 * np.ones(nu)
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 2 * Q).set_constraints(umin, umax)
    U = pl.solve(x0).cpu().numpy().reshape(batch, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    obj = pl.info.obj.cpu().numpy()
    H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=2 * Q))
    nact = 0
    for i in range(0, batch, 7):
        s = solve_box_qp(H, F @ x0[i] + f0, np.tile(umin, N), np.tile(umax, N))
        assert np.abs(U[i] - s["x"]).max() <= PRIMAL_RTOL * max(1.0, np.abs(s["x"]).max()), i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * max(abs(s["primal objective"]), 1e-12), i
        nact += int(((np.abs(s["x"] - np.tile(umin, N)) < 1e-9) | (np.abs(s["x"] - np.tile(umax, N)) < 1e-9)).sum())
    assert nact > 0


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_and_loads_satisfy_kkt_everywhere(seed):
    """Random (nx, nu, N) with n <= 128, random bound widths and initial-state scales, 700 instances each: small batches take
    the single 32-slot pass and, with tight bounds, overflow it into the 64-row pass and the any-size kernel.  Every instance
    must carry the optimality certificate."""
    import torch
    from mpc_amd.planner import Planner
    rng = np.random.default_rng(1000 + seed)
    nu = int(rng.integers(1, 5)); nx = int(rng.integers(2, 13))
    N = int(rng.integers(3, max(4, 128 // nu) + 1))
    batch = 700
    A = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    A *= min(1.0, 1.02 / np.abs(np.linalg.eigvals(A)).max())      # at most mildly unstable: H stays well conditioned in fp64
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.02, 0.5, nu))
    w = float(rng.choice([0.02, 0.1, 0.5]))
    umin, umax = -w * rng.uniform(0.5, 1.5, nu), w * rng.uniform(0.5, 1.5, nu)
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 3 * Q).set_constraints(umin, umax)
    x0 = torch.as_tensor(rng.uniform(-1, 1, (batch, nx)) * float(rng.choice([0.3, 1.0, 3.0])), device="cuda:0")
    U = pl.solve(x0).reshape(batch, -1)
    _assert_kkt_everywhere(pl, x0, U, umin, umax, N)


def test_hopelessly_conditioned_model_is_never_reported_solved_when_it_is_not():
    """An unstable model over a long horizon (spectral radius 1.3, N = 109: cond(H) ~ 1e14, the explicit inverse has two
    digits) with nearly every input saturated.  The solve refines with Hs itself and must then CHECK the refined point: U
    stays inside the bounds for every instance, and an instance is reported SOLVED only if its point is feasible and its
    multipliers have the right sign (found by the randomised test above: 10 of 700 instances came back SOLVED with bound
    violations up to 0.16 before the check existed)."""
    import torch
    from mpc_amd.planner import Planner
    rng = np.random.default_rng(1000)
    nu = int(rng.integers(1, 5)); nx = int(rng.integers(2, 13))
    N = int(rng.integers(3, max(4, 128 // nu) + 1))
    batch = 700
    A = np.eye(nx) + 0.15 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.diag(rng.uniform(0.5, 2.0, nx)); R = np.diag(rng.uniform(0.02, 0.5, nu))
    w = float(rng.choice([0.02, 0.1, 0.5]))
    umin, umax = -w * rng.uniform(0.5, 1.5, nu), w * rng.uniform(0.5, 1.5, nu)
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 3 * Q).set_constraints(umin, umax)
    x0 = torch.as_tensor(rng.uniform(-1, 1, (batch, nx)) * float(rng.choice([0.3, 1.0, 3.0])), device="cuda:0")
    U = pl.solve(x0, want_multipliers=True).reshape(batch, -1)
    assert (nx, nu, N) == (7, 1, 109) and np.linalg.cond(pl.get("H").cpu().numpy()) > 1e12
    lb = torch.as_tensor(np.tile(umin, N), device="cuda:0"); ub = torch.as_tensor(np.tile(umax, N), device="cuda:0")
    assert bool((U <= ub + 1e-12).all()) and bool((U >= lb - 1e-12).all())
    st = pl.info.status
    assert bool((st <= 1).all())                      # solved, or honestly "inaccurate"
    lam = pl.info.lam[st == 0]; Us = U[st == 0]
    at_ub = Us >= ub - 1e-12; at_lb = Us <= lb + 1e-12
    assert bool((lam[at_ub & ~at_lb] >= -1e-6 * lam.abs().max()).all())
    assert bool((lam[at_lb & ~at_ub] <= 1e-6 * lam.abs().max()).all())
    assert int((st == 0).sum()) > batch // 2


@pytest.mark.parametrize("batch,max_iter", [(65, 0), (9000, 0), (65, 5)])
def test_one_sided_input_bounds_match_oracle(batch, max_iter):
    """u_min = -inf on one input and u_max = +inf on another: rows bounded on one side only.  (Found by tools/fuzz_edges.py:
    the violation tolerance of a row was scaled with max(|lo|, |hi|), so an infinite side made the finite one untestable and
    the bound was silently ignored.)"""
    from mpc_amd.planner import Planner
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    rng = np.random.default_rng(8)
    nx, nu, N = 6, 3, 12
    A = 0.9 * np.eye(nx) + 0.1 * rng.standard_normal((nx, nx)) / np.sqrt(nx)
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx) + 0.1
    Q = np.eye(nx); R = 0.1 * np.eye(nu)
    umin = np.array([-np.inf, -0.3, -0.2]); umax = np.array([0.4, 0.25, np.inf])
    x0 = rng.uniform(-2, 2, (batch, nx))
    pl = Planner(nx, nu, N).set_model(A, B).set_cost(Q, R, 2 * Q).set_constraints(umin, umax)
    U = pl.solve(x0, max_iter=max_iter).cpu().numpy().reshape(batch, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    obj = pl.info.obj.cpu().numpy()
    H, F, f0 = mo.condense(dict(N=N, nx=nx, nu=nu, A=A, B=B, Q=Q, R=R, P=2 * Q))
    lb, ub = np.tile(umin, N), np.tile(umax, N)
    assert (U <= ub + 1e-12).all() and (U >= lb - 1e-12).all()
    nbind = 0
    for i in range(0, batch, max(1, batch // 12)):
        s = solve_box_qp(H, F @ x0[i] + f0, lb, ub)
        assert np.abs(U[i] - s["x"]).max() <= PRIMAL_RTOL * max(1.0, np.abs(s["x"]).max()), i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * max(abs(s["primal objective"]), 1e-12), i
        nbind += int((np.abs(s["x"] - ub) < 1e-9).sum() + (np.abs(s["x"] - lb) < 1e-9).sum())
    assert nbind > 0


@pytest.mark.parametrize("cfg,batch,split", [(3, 5000, True), (3, 20000, False), (2, 3000, True)])
def test_a_recorded_solve_replays_bit_for_bit(cfg, batch, split):
    """Planner.capture_solve: the launch chain of one solve as a HIP graph -- component-split cascade, the whole-instance
    cascade with its cold-start ordering, the one-pass small batch -- replayed on new x0, twice in a row, equals the eager
    solves bit for bit (every solve leaves the device-side counters ready for the next, so the chain carries no host state)."""
    import torch
    from mpc_amd.configs import make_config
    prob = make_config(cfg, batch=batch, seed=12)
    pl = _planner(prob, split)
    xa = torch.as_tensor(prob["x0"], device="cuda:0")
    xb = torch.as_tensor(1.3 * prob["x0"][::-1].copy(), device="cuda:0")
    Ua = pl.solve(xa).clone(); oa = pl.info.obj.clone()
    Ub = pl.solve(xb).clone(); ob = pl.info.obj.clone(); sb = pl.info.polish_steps.clone()
    g = pl.capture_solve(xa)
    for _ in range(2):
        U = g.replay(xb)
        torch.cuda.synchronize()
        assert torch.equal(U, Ub) and torch.equal(g.info.obj, ob) and torch.equal(g.info.polish_steps, sb)
        assert int((g.info.status != 0).sum()) == 0
        U = g.replay(xa)
        torch.cuda.synchronize()
        assert torch.equal(U, Ua) and torch.equal(g.info.obj, oa)
    # eager solves after the graph still work and agree
    assert torch.equal(pl.solve(xb), Ub)
    # while the graph lives, whatever would free the buffers its kernels point into is an error, not a fault at the next replay
    big = torch.cat([xa, xa, xa])[: 2 * batch + 64 * 1024]
    with pytest.raises(RuntimeError, match="recorded graph"):
        pl.solve(big)
    with pytest.raises(RuntimeError, match="recorded graph"):
        pl.set_cost(prob["Q"], 2.0 * prob["R"], prob["P"]).setup()
    assert torch.equal(g.replay(xb), Ub)
    g.close()
    with pytest.raises(RuntimeError, match="closed"):
        g.replay()
    pl.set_cost(prob["Q"], prob["R"], prob["P"]).setup()
    # (another batch size may take other kernels: the same point to rounding, not to the bit)
    assert float((pl.solve(big)[:batch] - Ua).abs().max()) <= 1e-11 * float(Ua.abs().max())


def test_rho_update_recovers_from_a_bad_penalty():
    """mpcqp_adapt_rho (BASELINE.json north_star, subsystem 2: "rho updates").  The ADMM stage alone (no active-set stage), started
    with a penalty a hundred times too small, runs into its iteration cap far from the answer; one update from the batch's
    final residuals (OSQP's ratio of the normalised residuals) brings rho back to the neighbourhood of the spectral choice and
    the same solve converges.  At the spectral choice itself the update leaves rho alone.  The kept scaled duals are
    rescaled with rho, so a warm start after the update confirms the solution.  (The reference has no counterpart: cvxopt's
    interior point has no penalty parameter; the answers are checked against the default solve, which meets the oracle bar.)"""
    import torch
    from mpc_amd.configs import make_config
    from mpc_amd.planner import Planner
    prob = make_config(3, batch=256, seed=1)
    pl = Planner(prob["nx"], prob["nu"], prob["N"]).set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"])
    pl.set_constraints(prob["u_min"], prob["u_max"])
    x0 = torch.as_tensor(prob["x0"], device="cuda:0")
    Uref = pl.solve(x0).clone()                                     # default: active-set kernels, oracle-accurate
    rho0 = pl.rho
    with pytest.raises(RuntimeError):
        pl.adapt_rho()                                              # no ADMM iterations on record
    err = lambda U: float((U - Uref).abs().max() / Uref.abs().max())
    # at the spectral choice: nothing to do
    pl.solve(x0, max_iter=2000, polish=False)
    it_default = float(pl.info.iters.float().mean())
    rho, changed = pl.adapt_rho()
    assert not changed and rho == rho0
    # a hundred times too small
    pl.setup(rho=rho0 * 0.01)
    U = pl.solve(x0, max_iter=2000, polish=False).clone()
    assert int(pl.info.iters.min()) == 2000 and err(U) > 1e-2       # stuck at the cap, far away
    rho, changed = pl.adapt_rho()
    assert changed and 0.2 * rho0 <= rho <= 5.0 * rho0 and pl.rho == rho
    U = pl.solve(x0, max_iter=2000, polish=False).clone()
    assert float(pl.info.iters.float().mean()) <= 1.5 * it_default and err(U) <= 1e-3
    # a hundred times too large: the update moves it down by more than the threshold
    pl.setup(rho=rho0 * 100.0)
    pl.solve(x0, max_iter=2000, polish=False, keep_state=True)
    rho, changed = pl.adapt_rho()
    assert changed and rho < rho0 * 100.0 / 5.0
    # the state kept across the update (w = y / rho rescaled): a warm start with the active-set stage lands on the answer
    Uw = pl.solve(x0, warm_start=True, max_iter=20, polish=True)
    assert bool((pl.info.status == 0).all()) and err(Uw) <= 1e-8
