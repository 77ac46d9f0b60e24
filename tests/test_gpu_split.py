"""Component split (mpc_amd/csrc/split.hip): a shared model whose condensed Hessian is block diagonal under a permutation --
independent axes, as the reference builds them (Launch3DPlanner.py:61-108: x, y, z rolled out separately) -- is solved
block by block.  Same bar as everywhere: primal 1e-6, objective 1e-8 relative against the certified oracle, and the
solver-independent KKT certificate for every instance."""
import numpy as np
import pytest
import scipy.linalg as sl

pytestmark = pytest.mark.gpu

PRIMAL_RTOL = 1e-6
OBJ_RTOL = 1e-8


def _block_problem(blocks, N, batch, seed, ubound=0.3, x0_amp=3.0, couple=0.0, stab=1.0):
    """Independent sub-systems (nx_i states, nu_i inputs each) stacked into one model; `couple` adds a weak coupling
    between the first two through the dynamics; `stab` < 1 pulls the poles in (long horizons stay well conditioned)."""
    rng = np.random.default_rng(seed)
    As, Bs = [], []
    for nx_i, nu_i in blocks:
        As.append(stab * np.eye(nx_i) + 0.15 * rng.standard_normal((nx_i, nx_i)) / np.sqrt(nx_i))
        Bs.append(rng.standard_normal((nx_i, nu_i)) / np.sqrt(nx_i))
    A = sl.block_diag(*As); B = sl.block_diag(*Bs)
    if couple:
        A[0, blocks[0][0]] = couple
    nx, nu = A.shape[0], B.shape[1]
    return dict(nx=nx, nu=nu, N=N, A=A, B=B, Q=np.eye(nx), R=0.05 * np.eye(nu), P=2.0 * np.eye(nx),
                u_min=-ubound * np.ones(nu), u_max=0.8 * ubound * np.ones(nu), x0=rng.uniform(-x0_amp, x0_amp, (batch, nx)))


def _planner(prob, split=True):
    from mpc_amd.planner import Planner
    pl = Planner(prob["nx"], prob["nu"], prob["N"])
    pl.split = split
    return pl.set_model(prob["A"], prob["B"]).set_cost(prob["Q"], prob["R"], prob["P"]).set_constraints(prob["u_min"], prob["u_max"])


def _check_against_oracle(pl, prob, U, every):
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    obj = pl.info.obj.cpu().numpy()
    for i in range(0, U.shape[0], every):
        s = solve_box_qp(H, F @ prob["x0"][i] + f0, lb, ub)
        assert s["kkt"]["max"] <= 1e-10
        assert np.abs(U[i] - s["x"]).max() <= PRIMAL_RTOL * np.abs(s["x"]).max(), i
        assert abs(obj[i] - s["primal objective"]) <= OBJ_RTOL * abs(s["primal objective"]), i
    # every instance: the certificate from H itself
    q = prob["x0"] @ F.T + f0
    G = U @ H + q
    assert (U <= ub + 1e-12).all() and (U >= lb - 1e-12).all()
    free = (U < ub - 1e-9) & (U > lb + 1e-9)
    sg = np.maximum(np.abs(q).max(axis=1, keepdims=True), 1.0)
    assert (np.abs(G) * free <= 1e-8 * sg).all()
    assert ((G <= 1e-8 * sg) | (U < ub - 1e-9)).all() and ((G >= -1e-8 * sg) | (U > lb + 1e-9)).all()
    assert (np.abs(obj - (0.5 * (U * (U @ H)).sum(1) + (q * U).sum(1))) <= 1e-8 * np.maximum(np.abs(obj), 1e-9)).all()
    return free


def test_the_quadrotor_is_four_chains_and_the_cart_pole_is_one():
    from mpc_amd.configs import make_config
    quad = _planner(make_config(3, batch=8, seed=0)).setup()
    info = quad.split_info()
    assert info["components"] == 4 and info["rows"] == 30
    assert info["row_map"] == [r % 4 for r in range(120)]         # thrust / roll / pitch / yaw: one input channel each
    assert _planner(make_config(2, batch=8, seed=0)).setup().split_info()["components"] == 0
    assert _planner(make_config(1, batch=8, seed=0)).setup().split_info()["components"] == 0
    off = _planner(make_config(3, batch=8, seed=0), split=False).setup()
    assert off.split_info()["components"] == 0


@pytest.mark.parametrize("blocks,N,rows", [([(2, 1), (3, 2), (2, 1)], 12, 24),       # three components of 12, 24, 12 rows
                                           ([(2, 1)] * 5, 9, 10),                    # five small ones (rows padded to 16 in LDS)
                                           ([(3, 1), (2, 1)], 50, 50),               # two of 50 rows: the 64-row layouts
                                           ([(2, 1), (2, 1), (1, 1), (3, 1), (2, 1), (2, 1), (1, 1), (2, 1)], 7, 8)])   # eight
def test_block_structures_match_oracle(blocks, N, rows):
    prob = _block_problem(blocks, N, batch=600, seed=len(blocks) + N)
    pl = _planner(prob)
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(600, -1)
    info = pl.split_info()
    assert info["components"] == len(blocks) and info["rows"] == rows
    assert (pl.info.status.cpu().numpy() == 0).all()
    free = _check_against_oracle(pl, prob, U, every=60)
    assert (~free).sum() > 600                                   # the bounds do bind
    # the same batch unsplit: the two paths land on the same point
    pu = _planner(prob, split=False)
    Uu = pu.solve(prob["x0"]).cpu().numpy().reshape(600, -1)
    assert np.abs(U - Uu).max() <= 1e-9 * np.abs(Uu).max()
    # same steps, component by component (compared where the whole-instance kernels report them all: an instance that ends in
    # their any-size kernel reports that kernel's steps only)
    st, su = pl.info.polish_steps.cpu().numpy(), pu.info.polish_steps.cpu().numpy()
    assert np.array_equal(st[st <= 16], su[st <= 16])
    assert np.abs(pl.info.obj.cpu().numpy() - pu.info.obj.cpu().numpy()).max() <= 1e-10 * np.abs(pu.info.obj.cpu().numpy()).max()


def test_a_component_with_more_active_rows_than_the_last_pass_holds_is_redone_as_a_whole():
    """Two chains of 50 rows under bounds so tight that a third of the components bind on more than 32 rows: beyond the split
    cascade's last pass, so those instances are solved again unsplit (split_finish_kernel's queue) -- and still to the bar.
    (cond(Hs) = 4.5e6, as configs[2].)"""
    prob = _block_problem([(3, 1), (2, 1)], 50, batch=300, seed=3, ubound=0.12, x0_amp=4.0, stab=0.9)
    pl = _planner(prob)
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(300, -1)
    assert pl.split_info()["components"] == 2
    assert (pl.info.status.cpu().numpy() == 0).all()
    free = _check_against_oracle(pl, prob, U, every=50)
    rm = np.array(pl.split_info()["row_map"])
    per = np.stack([((~free) & (rm == c)).sum(1) for c in range(2)], 1)
    assert (per > 32).sum() > 100 and (per <= 8).sum() > 40      # the redo path did run, and so did every pass before it


def test_weak_coupling_is_not_rounding_noise():
    """A coupling of 1e-7 between two chains is part of the problem, not of the rounding: no split at the default
    tolerance, and the answer is the coupled one."""
    prob = _block_problem([(2, 1), (2, 1), (2, 1)], 10, batch=64, seed=9, couple=1e-7)
    pl = _planner(prob)
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(64, -1)
    assert pl.split_info()["components"] in (0, 2)                # chains 0 and 1 stay together; chain 2 may stand alone
    _check_against_oracle(pl, prob, U, every=8)


@pytest.mark.parametrize("keep", [1, 0])
def test_state_is_kept_only_when_asked(keep):
    """keep_state = 0 skips the (z, w) writes of a cold solve; the answers are the same, shift() then has nothing to shift
    and a warm start falls back to a cold one.  With the state kept, a warm start from it (the kernels that take the
    instance as a whole read what the split kernels wrote) confirms the solution without a step."""
    import torch
    from mpc_amd.configs import make_config
    prob = make_config(3, batch=500, seed=4)
    pl = _planner(prob)
    pl.opts.keep_state = keep
    x0 = torch.as_tensor(prob["x0"], device="cuda:0")
    U = pl.solve(x0, want_multipliers=True).clone()
    lam = pl.info.lam.clone()
    assert pl.split_info()["components"] == 4 and bool((pl.info.status == 0).all())
    ref = _planner(prob, split=False)
    Ur = ref.solve(x0, want_multipliers=True)
    assert float((U - Ur).abs().max()) <= 1e-10 * float(Ur.abs().max())
    assert float((lam - ref.info.lam).abs().max()) <= 1e-9 * max(1.0, float(ref.info.lam.abs().max()))
    if keep:
        z = pl.get("z", batch=500); w = pl.get("w", batch=500)
        assert float((z - ref.get("z", batch=500)).abs().max()) <= 1e-10 * float(z.abs().max())
        assert float((w - ref.get("w", batch=500)).abs().max()) <= 1e-9 * max(1.0, float(w.abs().max()))
        Uw = pl.solve(x0, warm_start=True)
        assert float((Uw - U).abs().max()) <= 1e-10 * float(U.abs().max())
        assert int(pl.info.polish_steps.max()) == 0
        pl.shift()
    else:
        with pytest.raises(RuntimeError):
            pl.shift()
        Uw = pl.solve(x0, warm_start=True)                        # no state: a cold solve
        assert torch.equal(Uw, U)


def test_refactoring_with_other_weights_finds_the_split_again():
    """New weights couple nothing new, new bounds change the packed bounds: every factorisation packs its own image."""
    import torch
    from mpc_amd.configs import make_config
    prob = make_config(3, batch=256, seed=6)
    pl = _planner(prob)
    x0 = torch.as_tensor(prob["x0"], device="cuda:0")
    pl.solve(x0)
    prob2 = dict(prob); prob2["R"] = 0.3 * np.eye(4); prob2["u_max"] = np.array([4.0, 0.08, 0.05, 0.1])
    pl.set_cost(prob2["Q"], prob2["R"], prob2["P"]).set_constraints(prob2["u_min"], prob2["u_max"])
    U = pl.solve(x0).cpu().numpy().reshape(256, -1)
    assert pl.split_info()["components"] == 4
    _check_against_oracle(pl, prob2, U, every=64)


@pytest.mark.parametrize("batch", [300, 20000])
def test_one_sided_bounds_on_the_split_path(batch):
    """A channel without a lower bound and one without an upper bound, solved block by block.  The scan of the active-set
    passes packs a row index into the low bits of x - bound: an infinite bound there would make a NaN pattern of it (the
    staged bounds are kept finite for that reason; polish_quad.hip)."""
    prob = _block_problem([(3, 1), (3, 1), (4, 1)], N=16, batch=batch, seed=21, ubound=0.25, x0_amp=4.0, stab=0.9)
    prob["u_min"] = np.array([-np.inf, -0.25, -0.2]); prob["u_max"] = np.array([0.2, np.inf, 0.25])
    pl = _planner(prob)
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    assert pl.split_info()["components"] == 3
    assert (pl.info.status.cpu().numpy() == 0).all()
    free = _check_against_oracle(pl, prob, U, every=max(1, batch // 10))
    assert (~free).sum() > batch          # the finite sides do bind
    pu = _planner(prob, split=False)
    Uu = pu.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    assert np.abs(U - Uu).max() <= 1e-9 * max(1.0, np.abs(Uu).max())
