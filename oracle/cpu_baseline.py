"""ctypes wrapper of oracle/libqpipm.so (built from oracle/qp_ipm.c) -- TEST /
BASELINE INFRASTRUCTURE ONLY (see oracle/qp_oracle.py header for who may
import this)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libqpipm.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        lib.qp_ipm_batch_box.restype = C.c_int
        lib.qp_ipm_batch_box.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, C.c_int,
                                         C.c_double, C.c_double, C.c_double]
        lib.qp_ipm_solve.restype = C.c_int
        lib.qp_ipm_solve.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int, C.c_double, C.c_double,
                                     C.c_double, ip, dp]
        lib.qp_activeset_batch_box.restype = C.c_int
        lib.qp_activeset_batch_box.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, C.c_int]
        lib.qp_ipm_worksize.restype = C.c_long
        lib.qp_ipm_worksize.argtypes = [C.c_int, C.c_int]
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve_batch_box(H, F, f0, lb, ub, x0, nthreads=1, abstol=1e-7, reltol=1e-6, feastol=1e-7):
    """Interior-point solve of every row of x0; tolerances default to cvxopt's.
    Returns (U, iters, n_not_optimal, seconds)."""
    lib = load()
    H, F, f0, lb, ub, x0 = [np.ascontiguousarray(a, dtype=np.float64) for a in (H, F, f0, lb, ub, x0)]
    n, nx = F.shape
    B = x0.shape[0]
    U = np.zeros((B, n)); it = np.zeros(B, dtype=np.int32)
    t = time.perf_counter()
    bad = lib.qp_ipm_batch_box(n, nx, B, _p(H), _p(F), _p(f0), _p(lb), _p(ub), _p(x0), _p(U),
                               it.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads), abstol, reltol, feastol)
    return U, it, bad, time.perf_counter() - t


def solve_batch_box_activeset(H, F, f0, lb, ub, x0, nthreads=1):
    """The GPU's own algorithm on the host (oracle/qp_activeset.c): dual active-set method on the shared inverse, exact
    minimiser.  The inverse is formed once (NumPy), as the GPU's factor kernel does once per model and outside its timed
    solve.  Returns (U, steps, n_failed, seconds)."""
    lib = load()
    H, F, f0, lb, ub, x0 = [np.ascontiguousarray(a, dtype=np.float64) for a in (H, F, f0, lb, ub, x0)]
    n, nx = F.shape
    B = x0.shape[0]
    d = 1.0 / np.sqrt(np.diag(H))                   # Jacobi scaling before the inversion, as on the GPU
    Hs = H * d[:, None] * d[None, :]
    Hi = np.ascontiguousarray(np.linalg.inv(0.5 * (Hs + Hs.T)) * d[:, None] * d[None, :])
    Hi = np.ascontiguousarray(0.5 * (Hi + Hi.T))
    U = np.zeros((B, n)); st = np.zeros(B, dtype=np.int32)
    t = time.perf_counter()
    bad = lib.qp_activeset_batch_box(n, nx, B, _p(Hi), _p(F), _p(f0), _p(lb), _p(ub), _p(x0), _p(U),
                                     st.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    return U, st, bad, time.perf_counter() - t


def solve_batch_box_activeset_split(H, F, f0, lb, ub, x0, nthreads=1, tol=1e-14):
    """The same with the GPU's component split (mpc_amd/csrc/split.hip): rows that the Jacobi-scaled Hessian and its inverse
    do not couple (|entry| <= tol of the unit diagonal) form independent box QPs, solved block by block with the same C
    routine.  Returns (U, steps, n_failed, seconds, n_components); one component = the unsplit call."""
    from scipy.sparse.csgraph import connected_components
    lib = load()
    H, F, f0, lb, ub, x0 = [np.ascontiguousarray(a, dtype=np.float64) for a in (H, F, f0, lb, ub, x0)]
    n, nx = F.shape
    B = x0.shape[0]
    d = 1.0 / np.sqrt(np.diag(H))
    Hs = H * d[:, None] * d[None, :]
    His = np.linalg.inv(0.5 * (Hs + Hs.T)); His = 0.5 * (His + His.T)
    dg = np.sqrt(np.diag(His))
    ncomp, lab = connected_components((np.abs(Hs) > tol) | (np.abs(His) > tol * dg[:, None] * dg[None, :]), directed=False)
    Hi = His * d[:, None] * d[None, :]
    parts = []
    for c in range(ncomp):
        idx = np.where(lab == c)[0]
        parts.append((idx, np.ascontiguousarray(Hi[np.ix_(idx, idx)]), np.ascontiguousarray(F[idx]), np.ascontiguousarray(f0[idx]),
                      np.ascontiguousarray(lb[idx]), np.ascontiguousarray(ub[idx]), np.zeros((B, idx.size)), np.zeros(B, dtype=np.int32)))
    bad = 0
    t = time.perf_counter()
    for idx, Hc, Fc, fc, lc, uc, Uc, sc in parts:
        bad += lib.qp_activeset_batch_box(idx.size, nx, B, _p(Hc), _p(Fc), _p(fc), _p(lc), _p(uc), _p(x0), _p(Uc),
                                          sc.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    t = time.perf_counter() - t
    U = np.zeros((B, n)); st = np.zeros(B, dtype=np.int32)
    for idx, _, _, _, _, _, Uc, sc in parts:
        U[:, idx] = Uc; st += sc
    return U, st, bad, t, ncomp


def solve_dense(P, q, G, h, abstol=1e-7, reltol=1e-6, feastol=1e-7, maxiters=100):
    lib = load()
    P, q, G, h = [np.ascontiguousarray(a, dtype=np.float64) for a in (P, q.reshape(-1), G, h.reshape(-1))]
    n, m = q.size, h.size
    x = np.zeros(n); z = np.zeros(m); it = C.c_int(0)
    work = np.zeros(lib.qp_ipm_worksize(n, m))
    st = lib.qp_ipm_solve(n, m, _p(P), _p(q), _p(G), _p(h), _p(x), _p(z), maxiters, abstol, reltol, feastol,
                          C.byref(it), _p(work))
    return x, z, st, it.value
