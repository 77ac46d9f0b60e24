"""CPU tests of the host-side scenario physics of the closed-loop LTV path (mpc_amd/ltv.py): the bicycle's Jacobians
against finite differences.  (The QP path itself needs the GPU: tests/test_gpu_ltv.py.)"""
import numpy as np
import torch


def test_bicycle_jacobians_match_finite_differences():
    from mpc_amd.ltv import Bicycle
    m = Bicycle(L=2.7, dt=0.05)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((7, 4), generator=g, dtype=torch.float64) * 2 - 1
    x[:, 3] += 5.0
    u = (torch.rand((7, 2), generator=g, dtype=torch.float64) - 0.5) * torch.tensor([6.0, 1.0], dtype=torch.float64)
    A, B = m.jacobians(x, u)
    eps = 1e-6
    for j in range(4):
        dx = torch.zeros(4, dtype=torch.float64); dx[j] = eps
        fd = (m.step(x + dx, u) - m.step(x - dx, u)) / (2 * eps)
        assert torch.allclose(A[:, :, j], fd, atol=1e-8)
    for j in range(2):
        du = torch.zeros(2, dtype=torch.float64); du[j] = eps
        fd = (m.step(x, u + du) - m.step(x, u - du)) / (2 * eps)
        assert torch.allclose(B[:, :, j], fd, atol=1e-8)


def test_config_generators_are_seeded_and_shaped():
    from mpc_amd.configs import make_config
    for cfg, nx, nu in ((1, 2, 1), (2, 4, 1), (3, 12, 4), (4, 4, 2), (5, 20, 6)):
        a = make_config(cfg, batch=5, seed=3); b = make_config(cfg, batch=5, seed=3)
        assert a["nx"] == nx and a["nu"] == nu and a["x0"].shape == (5, nx)
        assert np.array_equal(a["x0"], b["x0"])
    p5 = make_config(5, batch=2)
    assert p5["C"].shape == (40, 20) and p5["N"] == 100
