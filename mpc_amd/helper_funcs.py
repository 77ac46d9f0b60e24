"""Host mirrors of the two helpers of the reference that sit on the QP path
(src/Planner/HelperFuncs.py:6-11 and :41-54), same names, arguments and return shapes.

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
    """Drop rows of [J | c] whose coefficients are all below `threshold` in magnitude
    (HelperFuncs.py:41-54).  Same messages as the reference."""
    J = np.asarray(J); c = np.asarray(c)
    null = (np.abs(J) < threshold).all(axis=1)
    if (null & (np.abs(c[:, 0]) > threshold)).any():
        for _ in range(int((null & (np.abs(c[:, 0]) > threshold)).sum())):
            print("Removing constraint with zero coefficient and non zero objective")
    print("Removing " + str(int(null.sum())) + " constraints")
    return J[~null], c[~null]
