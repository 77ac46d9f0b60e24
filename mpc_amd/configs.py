"""Synthetic linear-MPC problem families for BASELINE.json's configs.

The reference (AOE-khkhan/MPC) ships no state-space examples -- its nearest
default is the 1-D double integrator of
src/Planner/LaunchPlannerQPFormulation.py:14-44 -- so these generators fix the
concrete numbers SURVEY.md section 8(d) prescribes.  Host-side NumPy/SciPy only:
they produce *inputs* (model, weights, bounds, initial states); all solving
happens in the HIP library.

Every generator returns a dict with
  nx, nu, N, A (nx,nx), B (nx,nu), Q (nx,nx), R (nu,nu), P (nx,nx terminal),
  u_min (nu,), u_max (nu,), x0 (batch,nx), name
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sl


def _zoh(Ac, Bc, dt):
    nx, nu = Bc.shape
    M = np.zeros((nx + nu, nx + nu))
    M[:nx, :nx] = Ac * dt
    M[:nx, nx:] = Bc * dt
    E = sl.expm(M)
    return np.ascontiguousarray(E[:nx, :nx]), np.ascontiguousarray(E[:nx, nx:])


def double_integrator(batch=1, seed=0, N=20, dt=0.1):
    """Config 1: A=[[1,dt],[0,1]], B=[[dt^2/2],[dt]], Q=diag(1,0.1), R=0.01,
    |u|<=1, x0=(1,0); extra batch rows are x0 ~ U(-1,1)^2."""
    A = np.array([[1.0, dt], [0.0, 1.0]])
    B = np.array([[dt * dt / 2], [dt]])
    Q = np.diag([1.0, 0.1]); R = np.array([[0.01]])
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(-1, 1, (batch, 2)); x0[0] = [1.0, 0.0]
    return dict(name="double_integrator", nx=2, nu=1, N=N, A=A, B=B, Q=Q, R=R, P=Q.copy(),
                u_min=np.array([-1.0]), u_max=np.array([1.0]), x0=x0)


def cartpole(batch=4096, seed=0, N=50, dt=0.02):
    """Config 2: cart-pole linearised about upright, M=1, m=0.1, l=0.5, g=9.81,
    ZOH dt=0.02, Q=diag(1,1,10,1), R=0.1, |u|<=10, terminal P = DARE."""
    M, m, l, g = 1.0, 0.1, 0.5, 9.81
    Ac = np.zeros((4, 4)); Bc = np.zeros((4, 1))
    Ac[0, 1] = 1.0; Ac[2, 3] = 1.0
    Ac[1, 2] = -m * g / M
    Ac[3, 2] = (M + m) * g / (M * l)
    Bc[1, 0] = 1.0 / M
    Bc[3, 0] = -1.0 / (M * l)
    A, B = _zoh(Ac, Bc, dt)
    Q = np.diag([1.0, 1.0, 10.0, 1.0]); R = np.array([[0.1]])
    P = sl.solve_discrete_are(A, B, Q, R)
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(-0.2, 0.2, (batch, 4))
    return dict(name="cartpole", nx=4, nu=1, N=N, A=A, B=B, Q=Q, R=R, P=P,
                u_min=np.array([-10.0]), u_max=np.array([10.0]), x0=x0)


def quadrotor(batch=65536, seed=0, N=30, dt=0.05):
    """Config 3 (north star): quadrotor hover linearisation, 12 states
    (pos, vel, rpy, body rates), 4 inputs (thrust about hover + 3 torques),
    m=1 kg, I=diag(0.01,0.01,0.02), ZOH dt=0.05, Q=diag(10,10,10,1,1,1,10,10,10,1,1,1),
    R=0.1 I, terminal P = DARE, thrust +-5 N, torques +-0.1 N m."""
    nx, nu = 12, 4
    g, m = 9.81, 1.0
    J = np.diag([0.01, 0.01, 0.02])
    Ac = np.zeros((nx, nx)); Bc = np.zeros((nx, nu))
    Ac[0:3, 3:6] = np.eye(3)
    Ac[3, 7] = g          # x'' =  g * pitch
    Ac[4, 6] = -g         # y'' = -g * roll
    Ac[6:9, 9:12] = np.eye(3)
    Bc[5, 0] = 1.0 / m
    Bc[9:12, 1:4] = np.linalg.inv(J)
    A, B = _zoh(Ac, Bc, dt)
    Q = np.diag([10.0] * 3 + [1.0] * 3 + [10.0] * 3 + [1.0] * 3)
    R = 0.1 * np.eye(nu)
    P = sl.solve_discrete_are(A, B, Q, R)
    rng = np.random.default_rng(seed)
    x0 = np.concatenate([rng.uniform(-1, 1, (batch, 3)), rng.uniform(-0.5, 0.5, (batch, 3)),
                         rng.uniform(-0.2, 0.2, (batch, 3)), rng.uniform(-0.5, 0.5, (batch, 3))], axis=1)
    return dict(name="quadrotor", nx=nx, nu=nu, N=N, A=A, B=B, Q=Q, R=R, P=P,
                u_min=np.array([-5.0, -0.1, -0.1, -0.1]), u_max=np.array([5.0, 0.1, 0.1, 0.1]), x0=x0)


def bicycle(batch=8192, seed=0, N=20, dt=0.05):
    """Config 4: kinematic bicycle (x, y, psi, v; a, delta), L = 2.7 m, lane keeping at 5 m/s; the model is
    nonlinear, so there is no (A, B) here -- mpc_amd.ltv re-linearises it per instance and per step."""
    rng = np.random.default_rng(seed)
    x0 = np.stack((np.zeros(batch), rng.uniform(-2.0, 2.0, batch), rng.uniform(-0.3, 0.3, batch),
                   rng.uniform(3.0, 7.0, batch)), axis=1)
    return dict(name="bicycle", nx=4, nu=2, N=N, dt=dt, L=2.7, Q=np.diag([0.0, 1.0, 2.0, 0.5]), R=np.diag([0.1, 5.0]),
                P=np.diag([0.0, 5.0, 10.0, 2.5]), u_min=np.array([-3.0, -0.5]), u_max=np.array([3.0, 0.5]),
                v_ref=5.0, x0=x0)


def long_horizon(batch=256, seed=0, N=100, nx=20, nu=6):
    """Config 5: random stable LTI (A = 0.95 * orthogonal, perturbed), input box + state box on all states +
    2 nx random polytope rows per stage.  x0 is drawn small enough that the problem stays feasible."""
    rng = np.random.default_rng(seed)
    Qo, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    A = 0.95 * Qo + 0.02 * rng.standard_normal((nx, nx))
    B = rng.standard_normal((nx, nu)) / np.sqrt(nx)
    Q = np.eye(nx); R = 0.1 * np.eye(nu)
    P = sl.solve_discrete_are(A, B, Q, R)
    C = rng.standard_normal((2 * nx, nx)) / np.sqrt(nx)
    x0 = rng.uniform(-0.6, 0.6, (batch, nx))
    return dict(name="long_horizon", nx=nx, nu=nu, N=N, A=A, B=B, Q=Q, R=R, P=P, u_min=-0.5 * np.ones(nu),
                u_max=0.5 * np.ones(nu), x_min=-1.2 * np.ones(nx), x_max=1.2 * np.ones(nx), C=C,
                c_min=-1.0 * np.ones(2 * nx), c_max=1.0 * np.ones(2 * nx), x0=x0)


def make_config(cfg, batch=None, seed=0):
    """BASELINE.json configs[cfg-1] (1-based as in SURVEY.md section 8d)."""
    gens = {1: (double_integrator, 1), 2: (cartpole, 4096), 3: (quadrotor, 65536), 4: (bicycle, 8192), 5: (long_horizon, 256)}
    if cfg not in gens:
        raise ValueError(f"config {cfg} not available")
    fn, nb = gens[cfg]
    return fn(batch=nb if batch is None else batch, seed=seed)
