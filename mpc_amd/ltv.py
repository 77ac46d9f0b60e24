"""Closed-loop LTV MPC by re-linearisation about the shifted previous plan (BASELINE.json config 4:
kinematic bicycle, warm-start shifting, many parallel rollouts).

Reference counterpart: the sequential-QP loops that re-pose the QP around the previous answer and add the
increment back (`D = D + optX`, CollocationBasedPlanner2D.py:85-365; `f += delf`, SE3Planner.py:103-223).
Here every rollout has its OWN linearisation (one model per instance), all of them are assembled, factored
and solved by the HIP library in one call each; this module only holds the scenario physics (the nonlinear
model and its Jacobians, batched torch ops on the device) -- the part the reference keeps in its scripts too.
"""
from __future__ import annotations

import torch

from .planner import Planner


class Bicycle:
    """Kinematic bicycle, state (x, y, psi, v), input (a, delta), explicit Euler with step dt."""
    nx, nu = 4, 2

    def __init__(self, L=2.7, dt=0.05):
        self.L, self.dt = float(L), float(dt)

    def step(self, x, u):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float64 and u.dtype == torch.float64:
            # the plant of the closed loop in one launch (the tensor form below is a dozen element-wise kernels)
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            x, u = x.contiguous(), u.contiguous()
            out = torch.empty_like(x)
            p = lambda t: C.c_void_p(t.data_ptr())
            rc = lib.mpcqp_ltv_bicycle_step(x.device.index or 0, x.shape[0], self.dt, self.L, p(x), p(u), p(out),
                                            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"mpcqp_ltv_bicycle_step failed with code {rc}")
            return out
        px, py, psi, v = x.unbind(-1)
        a, delta = u.unbind(-1)
        dt, L = self.dt, self.L
        return torch.stack((px + dt * v * torch.cos(psi), py + dt * v * torch.sin(psi),
                            psi + dt * v / L * torch.tan(delta), v + dt * a), dim=-1)

    def jacobians(self, x, u):
        """A = df/dx (..., 4, 4), B = df/du (..., 4, 2) at (x, u)."""
        psi, v = x[..., 2], x[..., 3]
        delta = u[..., 1]
        dt, L = self.dt, self.L
        A = torch.eye(4, dtype=x.dtype, device=x.device).expand(x.shape[:-1] + (4, 4)).clone()
        A[..., 0, 2] = -dt * v * torch.sin(psi); A[..., 0, 3] = dt * torch.cos(psi)
        A[..., 1, 2] = dt * v * torch.cos(psi);  A[..., 1, 3] = dt * torch.sin(psi)
        A[..., 2, 3] = dt * torch.tan(delta) / L
        B = torch.zeros(x.shape[:-1] + (4, 2), dtype=x.dtype, device=x.device)
        B[..., 2, 1] = dt * v / (L * torch.cos(delta) ** 2)
        B[..., 3, 0] = dt
        return A, B


class ClosedLoopLTV:
    """Receding-horizon loop: nominal rollout of the shifted plan -> per-instance (A_k, B_k) -> QP in the input
    increments (x_ref - x_nominal, -u_nominal, bounds - u_nominal) -> apply the first input -> shift."""

    def __init__(self, model, N, Q, R, P, u_min, u_max, device="cuda:0"):
        self.model, self.N = model, int(N)
        self.device = torch.device(device)
        self.pl = Planner(model.nx, model.nu, self.N, device=self.device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.Q, self.R, self.P = (torch.as_tensor(m, **f64) for m in (Q, R, P))
        self.u_min = torch.as_tensor(u_min, **f64).reshape(1, 1, model.nu).contiguous()
        self.u_max = torch.as_tensor(u_max, **f64).reshape(1, 1, model.nu).contiguous()
        self.plan = None            # (B, N, nu) current input plan
        self.last = None            # tensors of the last QP, kept for inspection
        self._dx0 = None
        self._buf = None            # persistent linearisation buffers (bicycle kernel)
        self._registered = False
        self._plan2 = None          # the other half of the plan's ping-pong (fused apply / shift)

    def nominal(self, x, plan):
        xs = [x]
        for k in range(self.N):
            xs.append(self.model.step(xs[-1], plan[:, k]))
        return torch.stack(xs, dim=1)           # (B, N+1, nx)

    def linearize(self, x, ubar, x_ref):
        """nominal rollout, Jacobians and the deviation-form QP data: one HIP kernel for the bicycle, batched
        torch ops for any other model object with step()/jacobians()."""
        Bn, N, nx, nu = x.shape[0], self.N, self.model.nx, self.model.nu
        if isinstance(self.model, Bicycle):
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            f64 = dict(dtype=torch.float64, device=self.device)
            if self._buf is None or self._buf["A"].shape[0] != Bn:
                # persistent buffers: the handle remembers their addresses, so later steps only re-run the kernels
                self._buf = dict(xbar=torch.empty((Bn, N + 1, nx), **f64), A=torch.empty((Bn, N, nx, nx), **f64),
                                 B=torch.empty((Bn, N, nx, nu), **f64), xr=torch.empty((Bn, N, nx), **f64),
                                 ur=torch.empty((Bn, N, nu), **f64), lb=torch.empty((Bn, N, nu), **f64),
                                 ub=torch.empty((Bn, N, nu), **f64))
                self._registered = False
            out = self._buf
            p = lambda t: C.c_void_p(t.data_ptr())
            rc = lib.mpcqp_ltv_bicycle(self.device.index or 0, Bn, N, self.model.dt, self.model.L, p(x.contiguous()),
                                       p(ubar), p(x_ref.contiguous()), p(self.u_min), p(self.u_max), p(out["xbar"]),
                                       p(out["A"]), p(out["B"]), p(out["xr"]), p(out["ur"]), p(out["lb"]), p(out["ub"]),
                                       C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"mpcqp_ltv_bicycle failed with code {rc}")
            return out
        xbar = self.nominal(x, ubar)
        A, Bm = self.model.jacobians(xbar[:, :-1], ubar)
        return dict(xbar=xbar, A=A, B=Bm, xr=(x_ref - xbar[:, 1:]).contiguous(), ur=(-ubar).contiguous(),
                    lb=(self.u_min - ubar).contiguous(), ub=(self.u_max - ubar).contiguous())

    def capture(self, x, x_ref, warm=True, **opts):
        """Record one closed-loop step (linearise -> assemble -> factor -> solve -> apply -> shift) into a HIP graph.
        Call after at least two eager step()s (buffers registered, set-up calls asynchronous, workspaces grown).
        Afterwards ``replay(x, x_ref)`` runs the whole step with one graph launch and no host work in between."""
        if not (self._registered and self.pl.no_sync):
            raise RuntimeError("capture() needs two eager step() calls first")
        self._gx, self._gxr = x.clone(), x_ref.clone()
        self._gu0 = torch.empty((x.shape[0], self.model.nu), dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, capture_error_mode="relaxed"):
            self._step_in_place(self._gx, self._gxr, warm, opts)
        return self

    def replay(self, x, x_ref):
        """One recorded step on new data; returns the applied input (B,nu) (a persistent buffer: copy to keep)."""
        self._gx.copy_(x); self._gxr.copy_(x_ref)
        self._graph.replay()
        return self._gu0

    def _step_in_place(self, x, x_ref, warm, opts):
        # the same step as step() with every result in persistent buffers (graph replays read and write fixed addresses)
        ubar = self.plan
        self.linearize(x, ubar, x_ref)
        pl = self.pl
        pl.refresh()
        dU = pl.solve(self._dx0, warm_start=warm, **opts)
        plan = ubar + dU
        self._gu0.copy_(plan[:, 0])
        ubar[:, :-1].copy_(plan[:, 1:]); ubar[:, -1].copy_(plan[:, -1])
        pl.shift(recenter=True)

    def step(self, x, x_ref, warm=True, advance=False, **opts):
        """x (B,nx) current states, x_ref (B,N,nx) references for x_1..x_N.  Returns the applied input (B,nu); with
        ``advance=True`` (bicycle on the GPU) also the plant's next state under it, (u0, x_next) -- computed by the launch
        that applies the solution (mpcqp_ltv_bicycle_apply with x / x_next) instead of one of its own."""
        Bn = x.shape[0]
        first = self.plan is None
        if first:
            self.plan = torch.zeros((Bn, self.N, self.model.nu), dtype=torch.float64, device=self.device)
        ubar = self.plan
        lin = self.linearize(x, ubar, x_ref)
        pl = self.pl
        if lin is self._buf and self._registered:
            pl.refresh()        # same buffers, new contents: assemble + factor again, no pointer traffic
        else:
            pl.set_model(lin["A"], lin["B"])
            pl.set_cost(self.Q, self.R, self.P, x_ref=lin["xr"], u_ref=lin["ur"])
            pl.set_constraints(lin["lb"], lin["ub"])
            self._registered = lin is self._buf
        if self._dx0 is None or self._dx0.shape[0] != Bn:
            self._dx0 = torch.zeros((Bn, self.model.nx), dtype=torch.float64, device=self.device)
        dU = pl.solve(self._dx0, warm_start=(warm and not first), **opts)
        self.last = dict(xbar=lin["xbar"], ubar=ubar, A=lin["A"], B=lin["B"], x_ref=x_ref, dU=dU)
        # apply, then shift: drop the applied stage, repeat the last one; the kept multipliers move the same way and the
        # primal state is re-centred (the shifted plan is the origin of the next QP)
        if isinstance(self.model, Bicycle) and ubar.is_cuda:
            import ctypes as C
            if self._plan2 is None or self._plan2.shape != ubar.shape or self._plan2.data_ptr() == ubar.data_ptr():
                self._plan2 = torch.empty_like(ubar)
            u0 = torch.empty((Bn, self.model.nu), dtype=torch.float64, device=self.device)
            p = lambda t: C.c_void_p(t.data_ptr())
            x_next = None
            if advance:
                x = x.contiguous()
                x_next = torch.empty_like(x)
            # plan += dU, applied input, shifted plan, (the plant,) and the shift + re-centring of the kept state: one launch
            pl._check(pl._lib.mpcqp_ltv_bicycle_advance(pl._h, self.model.dt, self.model.L, p(ubar), p(dU.contiguous()),
                                                        p(self._plan2), p(u0), p(x) if advance else None,
                                                        p(x_next) if advance else None, pl._stream()))
            self.plan, self._plan2 = self._plan2, ubar       # `last["ubar"]` stays valid until the step after next
            pl.no_sync, pl.check_inputs = True, False
            return (u0, x_next) if advance else u0
        else:
            plan = ubar + dU
            u0 = plan[:, 0].clone()
            self.plan = torch.cat((plan[:, 1:], plan[:, -1:]), dim=1).contiguous()
            x_next = self.model.step(x, u0) if advance else None
        pl.shift(recenter=True)
        # after the first (synchronous, validated) step the set-up calls no longer wait for the device
        pl.no_sync, pl.check_inputs = True, False
        return (u0, x_next) if advance else u0
