"""GPU parity tests of the MPC hot path through the C-ABI (libmpcqp.so via ctypes).

Bar (BASELINE.json north star): primal within 1e-6 relative, objective within 1e-8
relative of the certified optimum, in fp64."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRIMAL_RTOL = 1e-6
OBJ_RTOL = 1e-8


def _planner(prob):
    from mpc_amd.planner import Planner
    pl = Planner(prob["nx"], prob["nu"], prob["N"])
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


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_solve_matches_golden(cfg, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mpc_cfg{cfg}.npz"))
    prob = _prob_from_golden(g)
    pl = _planner(prob)
    U = pl.solve(prob["x0"], want_multipliers=True).cpu().numpy().reshape(len(prob["x0"]), -1)
    info = pl.info
    assert (info.status.cpu().numpy() == 0).all()
    assert _rel(U, g["U"]).max() <= PRIMAL_RTOL
    obj = info.obj.cpu().numpy()
    # the device objective is in Jacobi-scaled variables, which leaves its value unchanged
    assert (np.abs(obj - g["objective"]) <= OBJ_RTOL * np.maximum(np.abs(g["objective"]), 1e-300)).all()
    lam = info.lam.cpu().numpy()
    assert ((lam != 0) == (g["lam"] != 0)).all(), "active set differs from the certified optimum"
    assert np.abs(lam - g["lam"]).max() <= 1e-6 * max(1.0, np.abs(g["lam"]).max())


@pytest.mark.parametrize("cfg,batch,scale", [(1, 64, 1.0), (2, 96, 8.0), (3, 200, 1.0), (3, 72, 3.0)])
def test_solve_matches_oracle_seeded(cfg, batch, scale):
    from mpc_amd.configs import make_config
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    prob = make_config(cfg, batch=batch, seed=7)
    prob["x0"] = prob["x0"] * scale
    pl = _planner(prob)
    U = pl.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    assert (pl.info.status.cpu().numpy() == 0).all()
    H, F, f0 = mo.condense(prob)
    lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
    obj = pl.info.obj.cpu().numpy()
    step = max(1, batch // 24)          # the oracle costs ~50 ms per instance at n = 120
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
