"""Planner facade over libmpcqp.so: model / cost / constraint set-up and
``solve(x0)`` returning the optimal input sequence for a whole batch of initial
states.

The reference has no such class -- every script under src/Planner/ is
module-level code that builds (P, q, G, h) with NumPy and calls
``cvxopt.solvers.qp`` once (LaunchPlannerQPFormulation.py:49-173).  The facade
is the entry-point set BASELINE.json's north star asks for; its semantics are
fixed in SURVEY.md section 8(b2).  What it replaces, step by step:

  set_model / set_cost   constants + prediction matrices + J'WJ
                         (LaunchPlannerQPFormulation.py:14-89, :130-162)
  set_constraints        input box rows (LaunchPlannerQPFormulation.py:94-107)
  solve(x0)              cvxopt.matrix + solvers.qp + soln['x'] (:165-174)

PyTorch is used only for device memory and streams; all arithmetic is in the
HIP library.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib


def _as_dev(x, device, shape=None, name="tensor"):
    if x is None:
        return None
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    t = t.to(device=device, dtype=torch.float64).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class SolveGraph:
    """A recorded solve (Planner.capture_solve).  ``replay(x0)`` copies the new initial states into the graph's input
    buffer, launches the graph and returns the output buffer U (B, N, nu) -- the SAME tensor every time, overwritten by
    the next replay; ``info`` likewise."""

    def __init__(self, planner, graph, x0, U, info):
        self.planner, self.graph, self.x0, self.U, self.info = planner, graph, x0, U, info
        # the recorded kernels hold pointers into the handle's workspaces: until close() the library refuses (with an error,
        # where a replay would otherwise fault) whatever would free or re-grow them -- a larger batch, a new factorisation
        planner._check(planner._lib.mpcqp_pin(planner._h, 1))
        self._pinned = True

    def close(self):
        """Drop the graph and let the planner re-factor / grow again."""
        if self._pinned:
            self._pinned = False
            self.graph = None
            if self.planner._h:
                self.planner._lib.mpcqp_pin(self.planner._h, 0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def replay(self, x0=None):
        if self.graph is None:
            raise RuntimeError("SolveGraph: closed")
        if x0 is not None:
            self.x0.copy_(x0)
        self.graph.replay()
        self.planner.info = self.info
        return self.U


class Planner:
    """Batched linear-MPC planner:  min sum 1/2|x_k-xr_k|_Q^2 + 1/2|u_k-ur_k|_R^2 + terminal P
    s.t. x_{k+1} = A_k x_k + B_k u_k, u_min <= u_k <= u_max, for a batch of x0."""

    def __init__(self, nx, nu, N, device="cuda:0"):
        self.nx, self.nu, self.N = int(nx), int(nu), int(N)
        self.n = self.N * self.nu
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("mpc_amd.Planner runs on a ROCm device only (no CPU path)")
        self._lib = _lib.load()
        self._h = None
        self._keep = {}          # tensors whose pointers the handle remembers
        self.reuse_outputs = False   # True: solve() returns the same U / info buffers for a given batch size on every call
        self._out = {}               # (they are overwritten by the next solve): no allocations, pre-built argument structs
        self._time_varying = None
        self._n_models = None
        self._dirty = True
        self.rho = None
        self.info = None
        self._general = None         # state-box / polytopic rows, if any
        self._general_path = "rows"  # "rows": Gram / constraint-space kernel; "dense": the general dense kernel
        self._rows_mem_limit = 32 << 30  # bytes of shared workspace (Gram matrix) above which "rows" falls back to "dense"
        self._gen_m, self._gen_kernel = 0, None     # rows and kernel of the general-constraint mode, set by setup()
        self._split = (True, 0.0)    # component split of a shared model (mpcqp_set_split): (enabled, coupling tolerance; 0 = default)
        self.no_sync = False         # True: set-up calls never wait for the device (closed loops)
        self.check_inputs = True     # False: skip the host-side sanity checks that synchronise (u_min <= u_max)
        self.opts = _lib.SolveOpts()
        self._lib.mpcqp_default_solve_opts(C.byref(self.opts))

    # kernel choice of the general-constraint mode: a change takes effect with the next set-up
    @property
    def general_path(self):
        return self._general_path

    @general_path.setter
    def general_path(self, v):
        if v not in ("rows", "dense"):
            raise ValueError("general_path must be 'rows' or 'dense'")
        self._general_path, self._dirty = v, True

    # component split: a shared model whose Hessian is block diagonal under a permutation (independent axes, as the
    # reference's Launch3DPlanner.py:61-108 builds x, y, z) is solved block by block; False switches the search off
    @property
    def split(self):
        return self._split[0]

    @split.setter
    def split(self, v):
        self._split, self._dirty = (bool(v), self._split[1]), True

    def split_info(self):
        """After setup(): {'components': C (0 = not split), 'rows': rows per component, 'row_map': component of every row}."""
        c, r = C.c_int(0), C.c_int(0)
        rm = (C.c_int * self.n)()
        self._check(self._lib.mpcqp_split_info(self._h, C.byref(c), C.byref(r), rm, self.n))
        return {"components": c.value, "rows": r.value, "row_map": list(rm)}

    @property
    def rows_mem_limit(self):
        return self._rows_mem_limit

    @rows_mem_limit.setter
    def rows_mem_limit(self, v):
        self._rows_mem_limit, self._dirty = int(v), True

    # ---- lifetime ------------------------------------------------------------------
    def _create(self, n_models, time_varying):
        if self._h is not None and (n_models, time_varying) == (self._n_models, self._time_varying):
            return
        self.close()
        dims = _lib.Dims(self.nx, self.nu, self.N, n_models, int(time_varying))
        h = C.c_void_p()
        rc = self._lib.mpcqp_create(C.byref(h), C.byref(dims), self.device.index or 0)
        _lib.check(self._lib, None, rc)
        self._h, self._n_models, self._time_varying = h, n_models, time_varying
        # re-register data given before the handle existed / for the old handle
        if "Q" in self._keep:
            self._push_cost()
        if "u_min" in self._keep:
            self._push_constraints()
        self._push_strides()

    def close(self):
        if self._h is not None:
            self._lib.mpcqp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _lib.check(self._lib, self._h, rc)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- problem data ----------------------------------------------------------------
    def set_model(self, A, B):
        """A: (nx,nx) | (N,nx,nx) | (b, N|1, nx, nx) with b = 1 (shared) or b = batch (one model per instance,
        e.g. an LTV re-linearisation about each instance's own trajectory);  B likewise with nu columns."""
        A = _as_dev(A, self.device); B = _as_dev(B, self.device)
        nx, nu, N = self.nx, self.nu, self.N
        nm = 1
        if A.dim() == 2:
            tv = False
        elif A.dim() == 3:
            tv = True
        elif A.dim() == 4:
            if B.dim() != 4 or B.shape[0] != A.shape[0] or B.shape[1] != A.shape[1]:
                raise ValueError("A and B must have the same leading (batch, stage) dimensions")
            nm = int(A.shape[0])
            tv = A.shape[1] != 1
            if tv and A.shape[1] != N:
                raise ValueError(f"A: stage dimension must be 1 or N={N}")
        else:
            raise ValueError("A must have 2, 3 or 4 dimensions")
        lead = (nm,) if nm > 1 else ()
        A = A.reshape(lead + ((N, nx, nx) if tv else (nx, nx))).contiguous()
        B = B.reshape(lead + ((N, nx, nu) if tv else (nx, nu))).contiguous()
        self._keep["A"], self._keep["B"] = A, B
        self._create(nm, tv)
        self._check(self._lib.mpcqp_set_model(self._h, _ptr(A), _ptr(B)))
        self._push_strides()
        self._dirty = True
        return self

    def _push_strides(self):
        if self._h is None:
            return
        k = self._keep
        if self._n_models == 1:
            if any(k.get(s, 0) for s in ("xref_stride", "uref_stride", "bnd_stride")):
                raise ValueError("per-instance references / bounds need a per-instance model (leading batch dimension)")
            return
        for name in ("x_ref", "u_ref", "u_min"):
            t = k.get(name)
            if t is not None and t.dim() == 3 and t.shape[0] != self._n_models:
                raise ValueError(f"{name}: leading dimension {t.shape[0]} != number of models {self._n_models}")
        self._check(self._lib.mpcqp_set_strides(self._h, k.get("xref_stride", 0), k.get("uref_stride", 0),
                                                k.get("bnd_stride", 0)))

    def _push_cost(self):
        k = self._keep
        self._check(self._lib.mpcqp_set_cost(self._h, _ptr(k["Q"]), _ptr(k["R"]), _ptr(k["P"]),
                                             _ptr(k.get("x_ref")), _ptr(k.get("u_ref"))))

    def set_cost(self, Q, R, P_terminal=None, x_ref=None, u_ref=None):
        """Q (nx,nx), R (nu,nu), P_terminal (nx,nx, default Q); x_ref (N,nx) for x_1..x_N, u_ref (N,nu);
        with one model per instance the references may carry a leading batch dimension."""
        nx, nu, N = self.nx, self.nu, self.N
        k = self._keep
        k["Q"] = _as_dev(Q, self.device, (nx, nx), "Q")
        k["R"] = _as_dev(R, self.device, (nu, nu), "R")
        k["P"] = _as_dev(Q if P_terminal is None else P_terminal, self.device, (nx, nx), "P_terminal")
        for name, val, cols in (("x_ref", x_ref, nx), ("u_ref", u_ref, nu)):
            t = _as_dev(val, self.device)
            stride = 0
            if t is not None:
                if t.dim() == 3:
                    if tuple(t.shape[1:]) != (N, cols):
                        raise ValueError(f"{name}: expected (batch, {N}, {cols}), got {tuple(t.shape)}")
                    stride = N * cols
                elif tuple(t.shape) != (N, cols):
                    raise ValueError(f"{name}: expected shape {(N, cols)}, got {tuple(t.shape)}")
            k[name] = t
            k["xref_stride" if name == "x_ref" else "uref_stride"] = stride
        if self._h is not None:
            self._push_cost()
            self._push_strides()
        self._dirty = True
        return self

    def _push_constraints(self):
        k = self._keep
        self._check(self._lib.mpcqp_set_constraints(self._h, _ptr(k["u_min"]), _ptr(k["u_max"]), k["per_stage"]))
        g = self._general or {}
        self._check(self._lib.mpcqp_set_state_constraints(
            self._h, _ptr(g.get("x_min")), _ptr(g.get("x_max")), _ptr(g.get("C")), _ptr(g.get("c_min")), _ptr(g.get("c_max")),
            int(g["C"].shape[1]) if "C" in g else 0))

    def set_constraints(self, u_min, u_max, x_min=None, x_max=None, C=None, c_min=None, c_max=None):
        """Input box u_min <= u_k <= u_max, (nu,) or (N,nu).  State box x_min <= x_k <= x_max and polytopic rows
        c_min <= C x_k <= c_max on x_1..x_N ((nx,)/(N,nx); C (nc,nx)/(N,nc,nx)) select the general-constraint path
        (reference: state rows = rows of the prediction matrix,
        LaunchPlannerQPFormulation.py:109-114, polytope rows Launch3DPlanner.py:134-142): the condensed QP then
        has one constraint matrix [I; Gamma rows; C Gamma rows] for the whole batch and goes to the dense kernel."""
        gen = {}
        if x_min is not None or x_max is not None:
            inf = torch.full((self.N, self.nx), float("inf"), dtype=torch.float64, device=self.device)
            lo = -inf if x_min is None else _as_dev(x_min, self.device).expand(self.N, self.nx)
            hi = inf if x_max is None else _as_dev(x_max, self.device).expand(self.N, self.nx)
            gen["x_min"], gen["x_max"] = lo.contiguous(), hi.contiguous()
        if C is not None:
            Cm = _as_dev(C, self.device)
            if Cm.dim() == 2:
                Cm = Cm.unsqueeze(0).expand(self.N, -1, -1)
            if Cm.dim() != 3 or Cm.shape[0] != self.N or Cm.shape[2] != self.nx:
                raise ValueError(f"C: expected (nc, {self.nx}) or ({self.N}, nc, {self.nx})")
            nc = Cm.shape[1]
            inf = torch.full((self.N, nc), float("inf"), dtype=torch.float64, device=self.device)
            gen["C"] = Cm.contiguous()
            gen["c_min"] = (-inf if c_min is None else _as_dev(c_min, self.device).expand(self.N, nc)).contiguous()
            gen["c_max"] = (inf if c_max is None else _as_dev(c_max, self.device).expand(self.N, nc)).contiguous()
        elif c_min is not None or c_max is not None:
            raise ValueError("c_min / c_max need C")
        self._general = gen or None
        u_min = _as_dev(u_min, self.device); u_max = _as_dev(u_max, self.device)
        if tuple(u_min.shape) != tuple(u_max.shape):
            raise ValueError("u_min and u_max must have the same shape")
        bnd_stride = 0
        if u_min.dim() == 3:                    # (batch, N, nu): one set of bounds per instance
            if tuple(u_min.shape[1:]) != (self.N, self.nu):
                raise ValueError(f"u_min/u_max: expected (batch, {self.N}, {self.nu})")
            per_stage, bnd_stride = 1, self.n
        elif u_min.numel() == self.nu:
            per_stage = 0
            u_min = u_min.reshape(self.nu); u_max = u_max.reshape(self.nu)
        elif u_min.numel() == self.n:
            per_stage = 1
            u_min = u_min.reshape(self.n); u_max = u_max.reshape(self.n)
        else:
            raise ValueError("u_min/u_max must have nu or N*nu elements (or a leading batch dimension)")
        if self.check_inputs and bool((u_min > u_max).any()):
            raise ValueError("u_min > u_max")
        self._keep.update(u_min=u_min.contiguous(), u_max=u_max.contiguous(), per_stage=per_stage, bnd_stride=bnd_stride)
        if self._h is not None:
            self._push_constraints()
            self._push_strides()
        self._dirty = True
        return self

    def refresh(self):
        """The registered model / cost / constraint buffers have new CONTENTS (same addresses): re-run assemble and
        factor before the next solve."""
        self._dirty = True
        return self

    # ---- kernels -----------------------------------------------------------------------
    def setup(self, rho=None, keep_state=False):
        """assemble (rollout, H, F, f0) + factor; called lazily by solve().  With ``self.no_sync`` (closed loops)
        the factor call does not wait for the device: a failed factorisation then shows up as per-instance status."""
        if self._h is None or "Q" not in self._keep or "u_min" not in self._keep:
            raise RuntimeError("set_model, set_cost and set_constraints must be called before solve")
        self._check(self._lib.mpcqp_set_split(self._h, int(self._split[0]), float(self._split[1])))
        self._check(self._lib.mpcqp_assemble(self._h, self._stream()))
        # state-box / polytopic rows (or N*nu > 128) are routed inside the library: mpcqp_factor builds the condensed rows on the
        # device and sets up the constraint-space or the dense kernel (general.hip)
        self._check(self._lib.mpcqp_set_general_opts(self._h, {"rows": 0, "dense": 1}[self.general_path],
                                                     int(self.rows_mem_limit)))
        general = self._general is not None or (self._n_models == 1 and self.n > 128)
        no_sync = bool(self.no_sync and (self._n_models > 1 or rho)) and not general
        fo = _lib.FactorOpts(float(rho) if rho else 0.0, int(no_sync), int(bool(keep_state)))
        self._check(self._lib.mpcqp_factor(self._h, C.byref(fo), self._stream()))
        m = C.c_int(0); rows = C.c_int(0)
        self._check(self._lib.mpcqp_general_rows(self._h, C.byref(m), C.byref(rows)))
        self._gen_m = int(m.value)
        self._gen_kernel = None if m.value == 0 else ("rows" if rows.value else "dense")
        if self._gen_m:
            self.rho = None
        else:
            self.rho = (1.0 if self._n_models > 1 else float(rho)) if no_sync else float(self.get("rho")[0])
        self._dirty = False
        return self

    def general_info(self):
        """Rows and kernel of the general-constraint mode after setup(): {'m': kept rows (0 = box kernels), 'kernel':
        'rows' | 'dense' | None}; the rows themselves are get('G'), get('E'), get('h0')."""
        return {"m": getattr(self, "_gen_m", 0), "kernel": getattr(self, "_gen_kernel", None)}

    def solve(self, x0, warm_start=None, want_multipliers=False, **opts):
        """x0 (B,nx) -> U (B,N,nu) on the device; per-instance diagnostics in ``self.info``
        (iters, status, obj, prim_res, dual_res, polish_steps[, lam]).
        opts: max_iter, min_iter, check_every, eps_abs, eps_rel, alpha, polish, polish_refine,
        polish_max_steps; rho (forces a re-factor)."""
        rho = opts.pop("rho", None)
        if self._dirty or (rho is not None and rho != self.rho):
            self.setup(rho, keep_state=bool(warm_start))
        x0 = _as_dev(x0, self.device)
        if x0.dim() == 1:
            x0 = x0.reshape(1, -1)
        if x0.dim() != 2 or x0.shape[1] != self.nx:
            raise ValueError(f"x0: expected (B, {self.nx}), got {tuple(x0.shape)}")
        B = x0.shape[0]
        gm = getattr(self, "_gen_m", 0)
        if self.reuse_outputs and not opts and warm_start is None:
            # serving loops: one set of output buffers and argument structs per batch size (an eager call otherwise spends
            # ~60 us of host time on ten small allocations and struct set-up, more than the launches themselves)
            rec = self._out.get((B, gm, want_multipliers))
            if rec is None:
                rec = self._out[(B, gm, want_multipliers)] = self._new_outputs(B, gm, want_multipliers)
            U, info, ci = rec
            self._check(self._lib.mpcqp_solve(self._h, _ptr(x0), B, _ptr(U), C.byref(ci), C.byref(self.opts), self._stream()))
            self.info = info
            return U
        o = _lib.SolveOpts.from_buffer_copy(self.opts)
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, v)
        if warm_start is not None:
            o.warm_start = int(bool(warm_start))
        U, info, ci = self._new_outputs(B, gm, want_multipliers)
        self._check(self._lib.mpcqp_solve(self._h, _ptr(x0), B, _ptr(U), C.byref(ci), C.byref(o), self._stream()))
        self.info = info
        return U

    def capture_solve(self, x0, want_multipliers=False, **opts):
        """Record ``solve(x0)`` (same options) into a HIP graph: returns a SolveGraph whose ``replay(x0_new)`` runs the
        whole launch chain of the solve with one graph launch and no host work per kernel.  Everything the chain needs
        is in place before the capture starts -- one eager solve grows the workspaces and sets the kernels' attributes,
        the output buffers are allocated outside the graph's pool -- and the chain itself does not depend on host state
        (every solve leaves its device-side counters ready for the next), so the recorded launches can be replayed any
        number of times.  The graph is tied to this batch size, these options and the current factorisation."""
        if self._dirty:
            self.setup()
        x0 = _as_dev(x0, self.device)
        if x0.dim() != 2 or x0.shape[1] != self.nx:
            raise ValueError(f"x0: expected (B, {self.nx}), got {tuple(x0.shape)}")
        xs = x0.clone()
        B = xs.shape[0]
        gm = getattr(self, "_gen_m", 0)
        o = _lib.SolveOpts.from_buffer_copy(self.opts)
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, v)
        U, info, ci = self._new_outputs(B, gm, want_multipliers)
        self._check(self._lib.mpcqp_solve(self._h, _ptr(xs), B, _ptr(U), C.byref(ci), C.byref(o), self._stream()))
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="relaxed"):
            self._check(self._lib.mpcqp_solve(self._h, _ptr(xs), B, _ptr(U), C.byref(ci), C.byref(o), self._stream()))
        return SolveGraph(self, graph, xs, U, info)

    def _new_outputs(self, B, gm, want_multipliers):
        dev = self.device
        U = torch.empty((B, self.N, self.nu), dtype=torch.float64, device=dev)
        info = SimpleNamespace(
            iters=torch.empty(B, dtype=torch.int32, device=dev),
            status=torch.empty(B, dtype=torch.int32, device=dev),
            obj=torch.empty(B, dtype=torch.float64, device=dev),
            prim_res=torch.empty(B, dtype=torch.float64, device=dev),
            dual_res=torch.empty(B, dtype=torch.float64, device=dev),
            polish_steps=torch.empty(B, dtype=torch.int32, device=dev),
            lam=torch.empty((B, gm if gm else self.n), dtype=torch.float64, device=dev) if want_multipliers else None)
        ci = _lib.Info(*[_ptr(getattr(info, f[0])) for f in _lib.Info._fields_])
        return U, info, ci

    def adapt_rho(self, threshold=0.0):
        """rho update of the ADMM stage (mpcqp_adapt_rho): after a solve that ran ADMM iterations, move the penalty by OSQP's
        ratio sqrt(r_prim / r_dual) of the batch's final residuals and factor again if that is more than `threshold`
        (default 5) away from the current one.  Returns (rho in force, whether it changed)."""
        rho, changed = C.c_double(0.0), C.c_int(0)
        self._check(self._lib.mpcqp_adapt_rho(self._h, float(threshold), C.byref(rho), C.byref(changed), self._stream()))
        if changed.value:
            self.rho = rho.value
        return rho.value, bool(changed.value)

    def shift(self, recenter=False):
        """Receding horizon: move the kept primal/dual state one stage forward (last stage repeated); recenter=True also
        zeroes the primal state in the same launch (shift().recenter() of a re-linearising loop)."""
        fn = self._lib.mpcqp_shift_recenter if recenter else self._lib.mpcqp_shift
        self._check(fn(self._h, self._stream()))
        return self

    def predict(self, x0, U):
        """States x_1..x_N (B, N, nx) that the inputs U (B, N, nu) produce from x0 (B, nx) under the model(s) set
        with set_model -- the reference's `x_traj = Gamma u* + ...` read-back (LaunchPlannerQPFormulation.py:176-183)."""
        if self._h is None:
            raise RuntimeError("set_model must be called before predict")
        x0 = _as_dev(x0, self.device); U = _as_dev(U, self.device)
        if x0.dim() == 1:
            x0 = x0.reshape(1, -1)
        B = x0.shape[0]
        if x0.shape[1] != self.nx or U.numel() != B * self.n:
            raise ValueError(f"predict: expected x0 (B, {self.nx}) and U (B, {self.N}, {self.nu})")
        X = torch.empty((B, self.N, self.nx), dtype=torch.float64, device=self.device)
        self._check(self._lib.mpcqp_predict(self._h, _ptr(x0), _ptr(U.reshape(B, -1).contiguous()), B, _ptr(X), self._stream()))
        return X

    def recenter(self):
        """Deviation-form re-linearisation: zero the kept primal state (the shifted plan becomes the origin of
        the next QP), keep the multipliers as the active-set guess."""
        self._check(self._lib.mpcqp_recenter(self._h, self._stream()))
        return self

    # ---- introspection -------------------------------------------------------------------
    _GET = {"H": 0, "F": 1, "f0": 2, "Gamma": 3, "Phi": 4, "Hs": 5, "M": 6, "Hi": 7, "d": 8, "rho": 9,
            "z": 10, "w": 11, "queues": 12, "G": 13, "E": 14, "h0": 15}

    def get(self, name, batch=None, model=0):
        n, p, nx = self.n, self.N * self.nx, self.nx
        shapes = {"H": (n, n), "F": (n, nx), "f0": (n,), "Gamma": (p, n), "Phi": (p, nx), "Hs": (n, n),
                  "M": (n, n), "Hi": (n, n), "d": (n,), "rho": (3,), "queues": (96,)}
        gm = getattr(self, "_gen_m", 0)
        shapes.update({"G": (gm, n), "E": (gm, nx), "h0": (gm,)})
        if name in ("z", "w"):
            if batch is None:
                raise ValueError("get('z'|'w') needs batch=")
            shape = (batch, n)
        else:
            shape = shapes[name]
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        self._check(self._lib.mpcqp_get(self._h, self._GET[name], int(model), _ptr(out), out.numel(), self._stream()))
        return out
