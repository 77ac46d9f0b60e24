"""CPU restatement of the MPC hot path -- TEST INFRASTRUCTURE ONLY (see
oracle/qp_oracle.py header for who may import this).

Three pieces, each the NumPy fp64 statement of what one HIP kernel computes:

condense()      horizon rollout into the prediction matrices and the condensed
                QP  H = Gamma' Qbar Gamma + Rbar,  q = F x0 + f0.
                Reference analogue: the double-integrator prediction matrices
                vi/xi (LaunchPlannerQPFormulation.py:49-89) and the least
                squares cost Q = J'WJ, f = -J'Wc (HelperFuncs.py:6-11,
                LaunchPlannerQPFormulation.py:151-162) with J = [Gamma; I],
                W = blkdiag(Qbar, Rbar).
factor()        Jacobi scaling, (Hs + rho I)^-1 and Hs^-1 and the affine maps
                from x0 to the ADMM offset / unconstrained minimiser.
admm_box()      the ADMM iteration the GPU runs for box constraints.
polish_box()    active-set KKT solve with pivoting repair (the step that
                brings an ADMM iterate to the exact minimiser).

The QP *answer* is certified by oracle/qp_oracle.py (KKT certificate); this
file exists so that GPU intermediates (H, F, iterates after k steps, polish
rounds) can be compared stage by stage.
"""
from __future__ import annotations

import numpy as np


def rollout_matrices(A, B, N):
    """Gamma (N nx, N nu) and Phi (N nx, nx) with X = Phi x0 + Gamma U,
    X = [x_1..x_N], U = [u_0..u_{N-1}].  A, B may be (nx,nx)/(nx,nu) (LTI) or
    (N,nx,nx)/(N,nx,nu) (LTV: x_{k+1} = A_k x_k + B_k u_k).

    Row block k is built from row block k-1 by one multiplication with A_k and
    a fresh B_k on the block diagonal -- the state-space form of the prefix
    sums at LaunchPlannerQPFormulation.py:68-70,:79-89."""
    A = np.asarray(A, float); B = np.asarray(B, float)
    if A.ndim == 2:
        A = np.broadcast_to(A, (N,) + A.shape)
    if B.ndim == 2:
        B = np.broadcast_to(B, (N,) + B.shape)
    nx, nu = B.shape[1:]
    Gam = np.zeros((N * nx, N * nu)); Phi = np.zeros((N * nx, nx))
    rowG = np.zeros((nx, N * nu)); rowP = np.eye(nx)
    for k in range(N):
        rowG = A[k] @ rowG
        rowG[:, k * nu:(k + 1) * nu] = B[k]
        rowP = A[k] @ rowP
        Gam[k * nx:(k + 1) * nx] = rowG
        Phi[k * nx:(k + 1) * nx] = rowP
    return Gam, Phi


def condense(prob, x_ref=None, u_ref=None):
    """Return (H, F, f0): 1/2 U'HU + (F x0 + f0)'U is the MPC cost
    sum_{k=1}^{N-1} |x_k - xr_k|^2_Q + |x_N - xr_N|^2_P + sum |u_k - ur_k|^2_R
    (each with the factor 1/2) up to a constant."""
    N, nx, nu = prob["N"], prob["nx"], prob["nu"]
    Gam, Phi = rollout_matrices(prob["A"], prob["B"], N)
    Qb = np.zeros((N * nx, N * nx)); Rb = np.zeros((N * nu, N * nu))
    for k in range(N):
        Qb[k * nx:(k + 1) * nx, k * nx:(k + 1) * nx] = prob["Q"] if k < N - 1 else prob["P"]
        Rb[k * nu:(k + 1) * nu, k * nu:(k + 1) * nu] = prob["R"]
    QG = Qb @ Gam
    H = Gam.T @ QG + Rb
    H = 0.5 * (H + H.T)
    F = QG.T @ Phi
    xr = np.zeros(N * nx) if x_ref is None else np.asarray(x_ref, float).reshape(-1)
    ur = np.zeros(N * nu) if u_ref is None else np.asarray(u_ref, float).reshape(-1)
    f0 = -(QG.T @ xr) - Rb @ ur
    return H, F, f0


def factor(H, F, f0, lb, ub, rho):
    """Shared (per model) quantities of the box-QP solver, all in the Jacobi
    scaled variables x = u / d,  d = 1/sqrt(diag H)."""
    d = 1.0 / np.sqrt(np.diag(H))
    Hs = H * d[:, None] * d[None, :]
    Hs = 0.5 * (Hs + Hs.T)
    n = H.shape[0]
    M = np.linalg.inv(Hs + rho * np.eye(n))
    Hi = np.linalg.inv(Hs)
    Fs = F * d[:, None]; f0s = f0 * d
    return dict(d=d, Hs=Hs, M=M, Hi=Hi, rho=rho, lbs=lb / d, ubs=ub / d,
                Kb=-(M @ Fs), kb=-(M @ f0s), Ku=-(Hi @ Fs), ku=-(Hi @ f0s), Fs=Fs, f0s=f0s)


def admm_box(fac, x0, iters, alpha=1.6, z0=None, w0=None, return_residuals=False):
    """`iters` ADMM iterations on  min 1/2 x'Hs x + qs'x, lbs<=x<=ubs  for every
    row of x0 (batch, nx).  State is (z, w) with w = y/rho the scaled dual.
    Iteration (v = z - w):
        x  = rho M v + b0,            b0 = -M qs = Kb x0 + kb
        xh = alpha x + (1-alpha) z
        z+ = clip(xh + w),  w+ = xh + w - z+
    Cold start: z = clip(x_unc), w = 0."""
    rho, M = fac["rho"], fac["M"]
    lbs, ubs = fac["lbs"][:, None], fac["ubs"][:, None]
    b0 = fac["Kb"] @ x0.T + fac["kb"][:, None]
    if z0 is None:
        xu = fac["Ku"] @ x0.T + fac["ku"][:, None]
        z = np.clip(xu, lbs, ubs); w = np.zeros_like(z)
    else:
        z = z0.T.copy(); w = w0.T.copy()
    rp = rd = None
    for _ in range(iters):
        x = rho * (M @ (z - w)) + b0
        xh = alpha * x + (1 - alpha) * z
        t = xh + w
        zn = np.clip(t, lbs, ubs)
        w = t - zn
        rp = np.abs(x - zn).max(0); rd = rho * np.abs(zn - z).max(0)
        z = zn
    if return_residuals:
        return z.T, w.T, rp, rd
    return z.T, w.T


def guess_side(fac, z, w):
    """Active-set guess handed to the polish: +1 upper, -1 lower, 0 free."""
    up = (z >= fac["ubs"][None, :]) & (w > 0)
    lo = (z <= fac["lbs"][None, :]) & (w < 0)
    return up.astype(np.int8) - lo.astype(np.int8)


def polish_box(fac, x0, side, max_rounds=50, refine=1):
    """Exact solve on a guessed active set with pivoting repair, one instance
    at a time (rows of x0 / side).  With S = Hi[A,A]:
        lam_A = S^-1 (x_unc[A] - bnd[A]),  x = x_unc - Hi[:,A] lam_A
    Constraints whose multiplier has the wrong sign are released, violated
    bounds are added; repeat until nothing changes.  Returns (x, lam, rounds)
    in scaled variables; rounds = -1 if max_rounds was hit."""
    Hi, Hs = fac["Hi"], fac["Hs"]
    lbs, ubs = fac["lbs"], fac["ubs"]
    nb, n = side.shape
    X = np.zeros((nb, n)); L = np.zeros((nb, n)); R = np.zeros(nb, int)
    for b in range(nb):
        qs = fac["Fs"] @ x0[b] + fac["f0s"]
        xu = fac["Ku"] @ x0[b] + fac["ku"]
        s = side[b].astype(int).copy()
        done = False
        for rnd in range(max_rounds):
            Aidx = np.nonzero(s)[0]
            bnd = np.where(s[Aidx] > 0, ubs[Aidx], lbs[Aidx])
            lam = np.zeros(n)
            x = xu.copy()
            if Aidx.size:
                la = np.linalg.solve(Hi[np.ix_(Aidx, Aidx)], xu[Aidx] - bnd)
                x = xu - Hi[:, Aidx] @ la
                x[Aidx] = bnd
                lam[Aidx] = la
                for _ in range(refine):
                    # residual of the true KKT system with Hs (not its inverse)
                    g = Hs @ x + qs + lam
                    g[Aidx] = 0.0
                    dx = -(Hi @ g)
                    dl = np.linalg.solve(Hi[np.ix_(Aidx, Aidx)], dx[Aidx])
                    dx = dx - Hi[:, Aidx] @ dl
                    dx[Aidx] = 0.0
                    x = x + dx; lam[Aidx] += dl
            else:
                for _ in range(refine):
                    x = x - Hi @ (Hs @ x + qs)
            tol = 1e-12
            bad = (s != 0) & (lam * s < -tol * max(1.0, np.abs(lam).max()))
            vu = (s == 0) & (x > ubs + tol * np.maximum(1.0, np.abs(ubs)))
            vl = (s == 0) & (x < lbs - tol * np.maximum(1.0, np.abs(lbs)))
            if not (bad.any() or vu.any() or vl.any()):
                done = True
                break
            s[bad] = 0; s[vu] = 1; s[vl] = -1
        X[b] = x; L[b] = lam; R[b] = rnd + 1 if done else -1
    return X, L, R
