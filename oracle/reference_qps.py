"""Restatement of the QP *data* the reference planner scripts assemble --
TEST INFRASTRUCTURE ONLY (see oracle/qp_oracle.py header for who may import).

Each function returns (P, q, G, h, A, b) exactly as the cited script hands
them to ``cvxopt.solvers.qp``.  tests/golden/make_golden.py checks every
function here against a capture of the real script's arguments (where the
script still runs on NumPy 2.x) -- so these are pinned to the reference's own
arithmetic, not just to my reading of it.

The scripts are module-level code with hard-coded constants; the functions
below take the same constants as keyword defaults so that tests can also build
larger instances (e.g. ``launch1d_qp(nodes=12)``) of the same family.
"""
from __future__ import annotations

import numpy as np


def cost_from_tasks(J, c, W):
    """Q = J'WJ, f = -J'Wc  (HelperFuncs.py:6-11)."""
    JtW = J.T @ W
    return JtW @ J, -(JtW @ c)


# ---------------------------------------------------------------------------
# Cubic-Hermite force spline -> double-integrator prediction matrices
# (LaunchPlannerQPFormulation.py:49-89; 3-axis version Launch3DPlanner.py:61-108)
# ---------------------------------------------------------------------------
def spline_prediction(nodes, dt, mass, naxis=1):
    """Return (vi, xi): velocity and position of every knot as linear maps of
    the decision vector [f_1, m_1, ..., f_{nodes-2}, m_{nodes-2}] (per axis,
    axis-interleaved as in Launch3DPlanner.py:61-62).

    Segment s (knot s -> s+1) adds  dt/(12 m) (6 f_s + m_s + 6 f_{s+1} - m_{s+1})
    to the velocity and  dt^2/(60 m) (21 f_s + 3 m_s + 9 f_{s+1} - 2 m_{s+1})
    + dt v_s to the position; knots 0 and nodes-1 are fixed, so only interior
    knots have columns.  Rows are then prefix-summed
    (LaunchPlannerQPFormulation.py:68-70, :79,:85,:89).
    """
    nvar = 2 * (nodes - 2) * naxis
    cv = dt / (12 * mass)
    cx = dt ** 2 / (60 * mass)
    vi = np.zeros((nodes * naxis, nvar))
    xs = np.zeros((nodes * naxis, nvar))          # per-segment position stencil
    for k in range(naxis):
        for s in range(nodes - 1):                # segment s ends at knot s+1
            row = (s + 1) * naxis + k
            if s >= 1:                            # start knot s is interior -> columns 2(s-1), 2(s-1)+1
                vi[row, (2 * s - 2) * naxis + k] = 6 * cv
                vi[row, (2 * s - 1) * naxis + k] = 1 * cv
                xs[row, (2 * s - 2) * naxis + k] = 21 * cx
                xs[row, (2 * s - 1) * naxis + k] = 3 * cx
            if s + 1 <= nodes - 2:                # end knot s+1 is interior
                vi[row, (2 * s + 0) * naxis + k] = 6 * cv
                vi[row, (2 * s + 1) * naxis + k] = -1 * cv
                xs[row, (2 * s + 0) * naxis + k] = 9 * cx
                xs[row, (2 * s + 1) * naxis + k] = -2 * cx
    for k in range(naxis):
        for i in range(1, nodes):
            vi[i * naxis + k] = vi[i * naxis + k] + vi[(i - 1) * naxis + k]
    xi = np.zeros_like(vi)
    for k in range(naxis):
        for i in range(nodes - 1):
            xi[(i + 1) * naxis + k] = xs[(i + 1) * naxis + k] + dt * vi[i * naxis + k] + xi[i * naxis + k]
    return vi, xi


def launch1d_qp(nodes=3, x0=0.65, v0=0.0, hDes=0.0, g=-9.81, mass=18, dt=0.1,
                mlim=2.0, xmax=0.82, xmin=0.5):
    """LaunchPlannerQPFormulation.py:14-168 (n = 2 nodes - 4, 8 ineq at nodes=3)."""
    vf = np.sqrt(-2.0 * g * hDes)
    fmin = mass * g
    fmax = mass * g - mass * g * 1.5
    f0 = m0 = ff = mf = 0.0
    xdes = np.array([[x0 - 0.15 * np.sin(2 * np.pi * i / (nodes - 1))] for i in range(nodes - 1)])
    vi, xi = spline_prediction(nodes, dt, mass)
    cv = dt / (12 * mass); cx = dt ** 2 / (60 * mass)
    delv0 = (6 * f0 + m0) * cv; delvf = (6 * ff - mf) * cv
    delx0 = (21 * f0 + 3 * m0) * cx; delxf = (9 * ff - 2 * mf) * cx
    n = 2 * nodes - 4
    # input box, four rows per interior knot (:94-107)
    Gf = np.zeros((2 * n, n)); hf = np.zeros((2 * n, 1))
    for i in range(nodes - 2):
        for r, (col, sgn, lim) in enumerate([(2 * i, 1, fmax), (2 * i, -1, -fmin),
                                             (2 * i + 1, 1, mlim), (2 * i + 1, -1, mlim)]):
            Gf[4 * i + r, col] = sgn
            hf[4 * i + r, 0] = lim
    # state box through the prediction rows (:109-118)
    hmax = np.ones((nodes - 1, 1)) * (xmax - x0 - delx0)
    hmin = -np.ones((nodes - 1, 1)) * (xmin - x0 - delx0)
    hmax[nodes - 2, 0] -= delxf
    hmin[nodes - 2, 0] += delxf
    for i in range(nodes - 1):
        drift = (i + 1) * dt * v0 + i * dt * delv0
        hmax[i, 0] -= drift
        hmin[i, 0] += drift
    G = np.vstack((Gf, xi[1:], -xi[1:])); h = np.vstack((hf, hmax, hmin))
    # cost: track positions, terminal velocity, regularise inputs (:130-162)
    c = xdes - x0
    for i in range(1, nodes):
        c[i - 1, 0] -= delx0 + i * dt * v0 + (i - 1) * dt * delv0
    c[nodes - 2, 0] += delxf
    J = np.vstack((xi[1:], vi[nodes - 1], np.identity(n)))
    cc = np.vstack((c, vf - v0 - delv0 - delvf, np.zeros((n, 1))))
    W = np.identity(J.shape[0])
    W[nodes:, nodes:] *= 1 / (10000 ** 2)
    P, q = cost_from_tasks(J, cc, W)
    return P, q, G, h, None, None


def collocation1d_qp(hgt=0.35, g=-9.81, T=1.0, v=0.2, dF=0.025):
    """CollocationBasedPlanner.py:7-57 (n=8, 10 ineq, 4 eq, P singular)."""
    om = -g / hgt
    T1, T2, T3 = T, T * T, T * T * T
    objJ = np.zeros((4, 8))
    objJ[0, 0] = 1.0
    objJ[1, :4] = [1.0, T1, T2, T3]
    objJ[2, 1] = 1.0
    objJ[3, 1:4] = [1.0, 2.0 * T1, 3.0 * T2]
    objC = np.array([[T * v / 2.0], [T * v / 2.0], [0.0], [0.0]])
    dyn = np.zeros((4, 8))
    dyn[0, 3], dyn[0, 7] = om, -om
    dyn[1, 2], dyn[1, 6] = om, -om
    dyn[2, 1], dyn[2, 3], dyn[2, 5] = -om, 6.0, om
    dyn[3, 0], dyn[3, 2], dyn[3, 4] = -om, 2.0, om
    cop = np.zeros((5, 8))
    for r, s in enumerate([0.0, 0.25, 0.5, 0.75, 1.0]):
        cop[r, 4:] = [1.0, (s ** 1) * T1, (s ** 2) * T2, (s ** 3) * T3]
    cop[0, 5:] = [0.0 * T1, 0.0 * T2, 0.0 * T3]
    P, q = cost_from_tasks(objJ, objC, np.identity(4))
    G = np.vstack((cop, -cop)); h = np.ones((10, 1)) * dF
    return P, q, G, h, dyn, np.zeros((4, 1))


def launch3d_qp(nodes=10, mu=0.6, hDes=0.15, mass=18, tLaunch=0.6):
    """Launch3DPlanner.py:14-199 (n = 6 (nodes-2) = 48, 150 ineq at nodes=10).
    The script itself raises at :33 on NumPy >= 1.24 (``g[zaxis]`` is a length-1
    array inside a list literal); the intent there is the scalar g_z, used here."""
    na, X, Y, Z = 3, 0, 1, 2
    x0 = np.array([[0.0], [0.0], [0.75]]); v0 = np.zeros((3, 1))
    g = np.array([[0.0], [0.0], [-9.81]])
    vf = np.array([[0.5], [0.0], [np.sqrt(-2.0 * g[Z, 0] * hDes)]])
    dt = tLaunch / nodes
    fmin = mass * g * 0
    fmax = -mass * g * 2.0 * (1 + mu)
    mmin = np.array([[-0.5], [-0.5], [-5.0]]) * dt
    mmax = np.array([[0.5], [0.5], [5.0]]) * dt
    xmax = np.array([[0.1], [0.1], [0.8]]); xmin = np.array([[-0.1], [-0.1], [0.5]])
    xdes = np.zeros((na * (nodes - 1), 1))
    for i in range(nodes - 1):
        xdes[na * i: na * (i + 1)] = x0 + np.array([[0.0], [0.0], [0.05]])
    f0 = -mass * g; m0 = np.zeros((3, 1)); ff = -mass * g * 0; mf = np.zeros((3, 1))
    vi, pi = spline_prediction(nodes, dt, mass, naxis=na)
    cv = dt / (12 * mass); cx = dt ** 2 / (60 * mass)
    delv0 = (6 * f0 + m0) * cv; delvf = (6 * ff - mf) * cv; delvg = dt * g
    delx0 = (21 * f0 + 3 * m0) * cx; delxf = (9 * ff - 2 * mf) * cx; delxg = dt ** 2 * g / 2
    nvar = na * 2 * (nodes - 2)
    fm = 2 + 2 * na
    Gf = np.zeros((fm * (nodes - 2), nvar)); hf = np.zeros((fm * (nodes - 2), 1))
    for i in range(nodes - 2):                                   # :118-131
        Gf[fm * i, 2 * i * na + Z] = 1 + mu;  hf[fm * i, 0] = fmax[Z, 0]
        Gf[fm * i + 1, 2 * i * na + Z] = -1;  hf[fm * i + 1, 0] = -fmin[Z, 0]
        for j in range(na):
            Gf[fm * i + 2 * j + 2, (2 * i + 1) * na + j] = 1;  hf[fm * i + 2 * j + 2, 0] = mmax[j, 0]
            Gf[fm * i + 2 * j + 3, (2 * i + 1) * na + j] = -1; hf[fm * i + 2 * j + 3, 0] = -mmin[j, 0]
    Gr = np.zeros((4 * (nodes - 2), nvar)); hr = np.zeros((4 * (nodes - 2), 1))
    for i in range(nodes - 2):                                   # friction pyramid :134-142
        for k in range(4):
            Gr[4 * i + k, na * 2 * i + X] = np.power(-1, int(k / 2))
            Gr[4 * i + k, na * 2 * i + Y] = np.power(-1, k)
            Gr[4 * i + k, na * 2 * i + Z] = -mu
    hmax = np.zeros((na * (nodes - 1), 1)); hmin = np.zeros((na * (nodes - 1), 1))
    for i in range(1, nodes):                                    # :149-156
        drift = x0 + delx0 + i * dt * v0 + (i - 1) * dt * delv0 + i * i * delxg
        hmax[na * (i - 1): na * i] = +(xmax - x0 - delx0 - i * dt * v0 - (i - 1) * dt * delv0 - i * i * delxg)
        hmin[na * (i - 1): na * i] = -(xmin - x0 - delx0 - i * dt * v0 - (i - 1) * dt * delv0 - i * i * delxg)
    hmax[na * (nodes - 2):] -= delxf
    hmin[na * (nodes - 2):] += delxf
    G = np.vstack((Gf, Gr, pi[na:], -pi[na:])); h = np.vstack((hf, hr, hmax, hmin))
    cp = xdes.copy()
    for i in range(1, nodes):                                    # :166-168
        cp[na * (i - 1): na * i] -= x0 + delx0 + i * dt * v0 + (i - 1) * dt * delv0 + (i * i) * delxg
    cp[na * (nodes - 2):] -= delxf
    Qp, fp = cost_from_tasks(pi[na:], cp, np.identity(na * (nodes - 1)) * 0.0)
    Qv, fv = cost_from_tasks(vi[(nodes - 1) * na: nodes * na],
                             vf - v0 - delv0 - delvf - (nodes - 1) * delvg, np.identity(na) * 1000)
    Wf = np.identity(nvar)
    for k in range(na):                                          # :182-190
        for i in range(nodes - 2):
            Wf[2 * i * na + k, 2 * i * na + k] *= (1 / 1000) if k != Z else (1 / 10000000)
            Wf[(2 * i + 1) * na + k, (2 * i + 1) * na + k] *= 1 / (1000 * dt * dt)
    Qf, ff_ = cost_from_tasks(np.identity(nvar), np.zeros((nvar, 1)), Wf)
    return Qp + Qv + Qf, fp + fv + ff_, G, h, None, None
