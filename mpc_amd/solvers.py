"""Drop-in for the part of ``cvxopt`` the reference planners use:

    from mpc_amd.solvers import matrix, qp, options          # instead of: from cvxopt import matrix, solvers
    soln = qp(P, q, G, h[, A, b]);  soln['x'], soln['status']

Reference call sites: CollocationBasedPlanner.py:52-60, LaunchPlannerQPFormulation.py:165-174,
Launch3DPlanner.py:196-202, CollocationBasedPlanner2D.py:16-20,359-362, SE3Planner.py:8,174-185.
The solve runs on the GPU in libmpcqp.so (``mpcqp_solve_dense_opts``); there is no CPU path.

Kept from cvxopt's contract:
  * ``matrix(x, tc='d')`` accepts 2-D arrays, the empty list ``[]`` (SE3Planner.py:152), flat lists (a column) and
    nested lists (cvxopt's convention: every inner list is a COLUMN); the result is indexable like the reference
    indexes it (``[i]``, ``[a:b]``, ``[i, j]``), converts with ``np.array(...)`` and can be the right operand of
    ``ndarray.dot`` / ``np.matmul``.
  * ``qp`` returns a dict with 'x', 'z', 'y', 's', 'status' ('optimal' | 'unknown'), 'primal objective',
    'dual objective', 'gap', 'relative gap', 'primal infeasibility', 'dual infeasibility', 'iterations';
    non-convergence is a status, errors are ordinary exceptions (ValueError for bad shapes, ArithmeticError when P
    is not positive semidefinite).
  * ``options`` is a process-global dict.  What each key the reference sets does here:
      show_progress  (SE3Planner.py:8)  one summary line per solve when truthy; default True, as in cvxopt
      maxiters       cap on the solver's iterations -- here active-set steps (cvxopt counts interior-point
                     iterations); reaching it gives status 'unknown'.  Unset: the kernel's own cap 20(n+m+p)+100
      refinement     refinement rounds on the final active set (cvxopt: iterative refinement of its KKT solves);
                     unset: 2
      abstol, reltol, feastol   the device solver is an exact active-set method and does not iterate towards them;
                     they GATE the status: 'optimal' only if the returned point meets cvxopt's stopping test
                     (primal / dual infeasibility <= feastol and gap <= abstol or relative gap <= reltol);
                     defaults are cvxopt's 1e-7, 1e-6, 1e-7
    ``initvals`` and ``kktsolver`` have no meaning for this method: passing them warns once per call.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib

options = {}


class matrix(np.ndarray):
    """Dense double matrix backed by NumPy (cvxopt.matrix look-alike for the operations the reference performs
    on it)."""

    def __new__(cls, x=None, size=None, tc="d"):
        if x is None or (isinstance(x, (list, tuple)) and len(x) == 0):
            arr = np.zeros((0, 1))
        else:
            nested = isinstance(x, (list, tuple)) and len(x) > 0 and isinstance(x[0], (list, tuple))
            arr = np.array(x, dtype=np.float64)
            if arr.ndim == 0:
                arr = arr.reshape(1, 1)
            elif arr.ndim == 1:
                arr = arr.reshape(-1, 1)          # cvxopt: a flat list is a column
            elif arr.ndim != 2:
                raise TypeError("matrix() needs a 1-D or 2-D array")
            elif nested:
                arr = arr.T                       # cvxopt: a list of lists is a list of COLUMNS
        if size is not None:
            arr = arr.reshape(size, order="F")
        return np.ascontiguousarray(arr).view(cls)

    @property
    def size(self):                                # cvxopt: (rows, cols)
        return self.shape

    def __repr__(self):                            # NumPy's own printing reads .size as a number
        return repr(np.asarray(self))

    __str__ = __repr__

    def __getitem__(self, idx):
        # cvxopt indexes a matrix with ONE index in column-major order and returns a column for slices
        if isinstance(idx, (int, np.integer)):
            return float(np.asarray(self).reshape(-1, order="F")[idx])
        if isinstance(idx, slice):
            return np.asarray(self).reshape(-1, order="F")[idx].reshape(-1, 1).view(matrix)
        out = np.asarray(self)[idx]
        return out.view(matrix) if isinstance(out, np.ndarray) and out.ndim == 2 else out


def _dev(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("mpc_amd.solvers runs on a ROCm device only (no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def qp_batch(P, q, G=None, h=None, A=None, b=None, device=None, max_steps=0, refine=None):
    """Batched form: q (B,n), h (B,m), b (B,p) per QP; P (B,n,n) | (n,n), G (B,m,n) | (m,n), A (B,p,n) | (p,n) per QP
    or ONE matrix shared by the whole batch (the MPC case: one Hessian, one constraint matrix, per-instance q and h).
    NumPy or torch.  Returns dict of torch tensors on the device: x (B,n), z (B,m), y (B,p), obj (B), status (B), iters (B)."""
    lib = _lib.load()
    dev = _default_device() if device is None else torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("mpc_amd.solvers runs on a ROCm device only (no CPU path)")
    to = lambda a: None if a is None else (a.to(device=dev, dtype=torch.float64).contiguous() if torch.is_tensor(a) else _dev(a, dev))
    P, q, G, h, A, b = map(to, (P, q, G, h, A, b))
    if q.dim() != 2:
        raise ValueError("q must be (B, n)")
    B, n = q.shape
    if P.dim() not in (2, 3) or tuple(P.shape[-2:]) != (n, n) or (P.dim() == 3 and P.shape[0] != B):
        raise ValueError("P must be (B, n, n) or (n, n)")
    m = 0 if G is None else G.shape[-2]
    p = 0 if A is None else A.shape[-2]
    if m and (G.dim() not in (2, 3) or G.shape[-1] != n or (G.dim() == 3 and G.shape[0] != B)
              or h is None or tuple(h.shape) != (B, m)):
        raise ValueError("G must be (B, m, n) or (m, n) and h (B, m)")
    if p and (A.dim() not in (2, 3) or A.shape[-1] != n or (A.dim() == 3 and A.shape[0] != B)
              or b is None or tuple(b.shape) != (B, p)):
        raise ValueError("A must be (B, p, n) or (p, n) and b (B, p)")
    x = torch.empty((B, n), dtype=torch.float64, device=dev)
    z = torch.zeros((B, max(m, 1)), dtype=torch.float64, device=dev)
    y = torch.zeros((B, max(p, 1)), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    ni = C.c_long(0)
    nd = lib.mpcqp_dense_workspace(n, m, p, C.byref(ni))
    ws = torch.empty(B * nd, dtype=torch.float64, device=dev)
    iws = torch.empty(B * ni.value, dtype=torch.int32, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    stride = lambda t, per: 0 if (t is None or t.dim() == 2) else per
    do = _lib.DenseOpts()
    lib.mpcqp_default_dense_opts(C.byref(do))
    do.max_steps = int(max_steps or 0)
    if refine is not None:
        do.refine = int(refine)
    rc = lib.mpcqp_solve_dense_opts(dev.index or 0, B, n, m, p, ptr(P), stride(P, n * n), ptr(q), n,
                                    ptr(G), stride(G, m * n), ptr(h), m, ptr(A), stride(A, p * n), ptr(b), p,
                                    ptr(x), ptr(z), ptr(y), ptr(obj), ptr(status), ptr(iters), ptr(ws), ptr(iws),
                                    C.byref(do), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc == _lib.EINVAL:
        raise ValueError("mpcqp_solve_dense: invalid argument")
    if rc != _lib.OK:
        raise RuntimeError(f"mpcqp_solve_dense failed with code {rc}")
    return dict(x=x, z=z[:, :m], y=y[:, :p], obj=obj, status=status, iters=iters)


def qp(P, q, G=None, h=None, A=None, b=None, solver=None, kktsolver=None, initvals=None, **kwargs):
    """cvxopt.solvers.qp look-alike (one QP).  min 1/2 x'Px + q'x  s.t.  Gx <= h, Ax = b."""
    ignored = [k for k, v in (("solver", solver), ("kktsolver", kktsolver), ("initvals", initvals)) if v is not None]
    ignored += sorted(kwargs)
    if ignored:
        warnings.warn("mpc_amd.solvers.qp ignores " + ", ".join(ignored) + ": the device solver is an active-set method "
                      "started from the empty set (no interior-point start, no KKT-solver hook)", stacklevel=2)
    Pn = np.asarray(P, dtype=np.float64)
    if Pn.ndim != 2 or Pn.shape[0] != Pn.shape[1]:
        raise ValueError("P must be a square matrix")
    n = Pn.shape[0]
    qn = np.asarray(q, dtype=np.float64).reshape(-1)
    if qn.size != n:
        raise ValueError("q must have n rows")

    def rows(M, v, name):
        if M is None or np.size(M) == 0:
            return None, None
        Mn = np.asarray(M, dtype=np.float64)
        vn = np.asarray(v, dtype=np.float64).reshape(-1)
        if Mn.ndim != 2 or Mn.shape[1] != n or vn.size != Mn.shape[0]:
            raise ValueError(f"{name}: inconsistent dimensions")
        return Mn, vn
    Gn, hn = rows(G, h, "G, h")
    An, bn = rows(A, b, "A, b")
    maxiters = options.get("maxiters")
    if maxiters is not None and (not isinstance(maxiters, (int, np.integer)) or maxiters < 1):
        raise ValueError("options['maxiters'] must be a positive integer")
    refinement = options.get("refinement")
    if refinement is not None and (not isinstance(refinement, (int, np.integer)) or refinement < 0):
        raise ValueError("options['refinement'] must be a nonnegative integer")
    tol = {}
    for key, default in (("abstol", 1e-7), ("reltol", 1e-6), ("feastol", 1e-7)):
        tol[key] = float(options.get(key, default))
    if tol["feastol"] <= 0.0:
        raise ValueError("options['feastol'] must be a positive scalar")
    bat = lambda a: None if a is None else a[None]
    r = qp_batch(Pn[None], qn[None], bat(Gn), bat(hn), bat(An), bat(bn), max_steps=maxiters or 0, refine=refinement)
    st = int(r["status"][0].item())
    if st == 3:
        raise ArithmeticError("P is not positive semidefinite (Cholesky of P + sigma I failed)")
    x = r["x"][0].cpu().numpy()
    z = r["z"][0].cpu().numpy(); y = r["y"][0].cpu().numpy()
    it = int(r["iters"][0].item())
    pobj = float(r["obj"][0].item())            # 1/2 x'Px + q'x as evaluated by the kernel
    # cvxopt's stopping quantities at the returned point (coneqp: resx / max(1,|q|), (resy, resz) / max(1,|(b,h)|), gap = s'z)
    s = (hn - Gn @ x) if Gn is not None else np.zeros(0)
    rx = Pn @ x + qn
    if Gn is not None:
        rx = rx + Gn.T @ z
    if An is not None:
        rx = rx + An.T @ y
    ry = (An @ x - bn) if An is not None else np.zeros(0)
    rz = np.minimum(s, 0.0)                     # the slack cvxopt carries is max(s, 0): what is left over is infeasibility
    dres = float(np.linalg.norm(rx) / max(1.0, np.linalg.norm(qn)))
    pres = float(max(np.linalg.norm(ry) / max(1.0, np.linalg.norm(bn) if bn is not None else 0.0),
                     np.linalg.norm(rz) / max(1.0, np.linalg.norm(hn) if hn is not None else 0.0)))
    gap = float(np.maximum(s, 0.0) @ z) if s.size else 0.0
    dobj = pobj - gap + float(y @ ry) if An is not None else pobj - gap
    if pobj < 0.0:
        relgap = gap / -pobj
    elif dobj > 0.0:
        relgap = gap / dobj
    else:
        relgap = None
    met = (pres <= tol["feastol"] and dres <= tol["feastol"]
           and (gap <= tol["abstol"] or (relgap is not None and relgap <= tol["reltol"])))
    status = "optimal" if (st == 0 and met) else "unknown"
    if options.get("show_progress", True):
        print(f"mpc_amd.solvers.qp: n={n} status={status} (device code {st}) active-set steps={it % 1000} "
              f"proximal iterations={it // 1000} pcost={pobj:.10g} gap={gap:.2e} pres={pres:.2e} dres={dres:.2e}")
    return {"x": matrix(x), "z": matrix(z), "y": matrix(y), "s": matrix(np.maximum(s, 0.0)),
            "status": status,
            "primal objective": pobj, "dual objective": dobj, "gap": gap, "relative gap": relgap,
            "primal infeasibility": pres, "dual infeasibility": dres,
            "primal slack": float(s.min()) if s.size else 0.0,
            "dual slack": float(z.min()) if z.size else 0.0, "iterations": it % 1000}
