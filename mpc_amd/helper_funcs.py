"""Host mirrors of the two helpers of the reference that sit on the QP path
(src/Planner/HelperFuncs.py:6-11 and :41-54), same names, arguments and return shapes.

``removeNullConstraints`` examines and compacts the rows on the device (rows_util.hip).
``getQuadProgCostFunction`` runs on the device when given device tensors (one GEMM pair through
torch's rocBLAS binding -- plumbing for callers that assemble their own J, c, W; the MPC path does
this contraction inside assemble_kernel) and in NumPy when given NumPy inputs, exactly as the
reference does."""
from __future__ import annotations

import numpy as np
import torch


def getQuadProgCostFunction(J, c, W):
    """Q = J'WJ, f = -J'Wc  (HelperFuncs.py:6-11); f has shape (n, 1)."""
    if torch.is_tensor(J):
        JtW = J.transpose(-1, -2) @ W
        return JtW @ J, -(JtW @ c)
    J = np.asarray(J); JtW = J.transpose().dot(W)
    return JtW.dot(J), -JtW.dot(c)


def removeNullConstraints(J, c, threshold):
    """Drop rows of [J | c] whose coefficients are all below `threshold` in magnitude (HelperFuncs.py:41-54; call site
    CollocationBasedPlanner2D.py:338).  Same messages and return shapes as the reference.  The rows are examined and
    compacted on the device (``mpcqp_remove_null_rows``); NumPy inputs are copied there and back, device tensors stay."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    as_numpy = not torch.is_tensor(J)
    if not torch.cuda.is_available():
        raise RuntimeError("mpc_amd.helper_funcs.removeNullConstraints runs on a ROCm device only (no CPU path)")
    dev = J.device if (not as_numpy and J.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    Jd = torch.as_tensor(np.asarray(J, dtype=np.float64) if as_numpy else J).to(device=dev, dtype=torch.float64).contiguous()
    cd = torch.as_tensor(np.asarray(c, dtype=np.float64) if as_numpy else c).to(device=dev, dtype=torch.float64).contiguous()
    if Jd.dim() != 2 or cd.numel() != Jd.shape[0]:
        raise ValueError("removeNullConstraints: J must be (rows, cols) and c (rows, 1)")
    rows, cols = Jd.shape
    Jo = torch.empty_like(Jd); co = torch.empty(rows, dtype=torch.float64, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    iws = torch.empty(max(2 * rows, 1), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.mpcqp_remove_null_rows(dev.index or 0, rows, cols, p(Jd), p(cd), float(threshold), p(Jo), p(co), p(counts),
                                    p(iws), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != _lib.OK:
        raise (ValueError if rc == _lib.EINVAL else RuntimeError)(f"mpcqp_remove_null_rows failed with code {rc}")
    kept, odd = (int(v) for v in counts.cpu())
    for _ in range(odd):
        print("Removing constraint with zero coefficient and non zero objective")
    print("Removing " + str(rows - kept) + " constraints")
    Jo, co = Jo[:kept], co[:kept].reshape(kept, 1)
    if as_numpy:
        return Jo.cpu().numpy(), co.cpu().numpy()
    return Jo, co
