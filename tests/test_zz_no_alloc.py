"""mpcqp_solve_dense_static (include/mpcqp.h): the dense solve with the workspace in a static device array, for callers
that cannot allocate.  Called through the C-ABI and compared with mpcqp_solve_dense (same kernel, caller's workspace) on
two calls in a row, the second with a smaller input; plus the size limit.  Collected last on purpose (file name)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(rng, B, n, m, p):
    M = rng.standard_normal((B, n, n))
    P = M @ M.transpose(0, 2, 1) / n + 1e-2 * np.eye(n)
    q = rng.standard_normal((B, n))
    G = rng.standard_normal((B, m, n))
    xf = rng.standard_normal((B, n))
    h = np.einsum("bmn,bn->bm", G, xf) + rng.uniform(0.05, 1.0, (B, m))
    A = rng.standard_normal((B, p, n))
    b = np.einsum("bpn,bn->bp", A, xf)
    return P, q, G, h, A, b


def _run(lib, static, P, q, G, h, A, b):
    import torch
    from mpc_amd import _lib
    dev = torch.device("cuda:0")
    B, n = q.shape
    m, p = G.shape[1], A.shape[1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    Pd, qd, Gd, hd, Ad, bd = map(t, (P, q, G, h, A, b))
    x = torch.full((B, n), float("nan"), dtype=torch.float64, device=dev)
    z = torch.full((B, max(m, 1)), float("nan"), dtype=torch.float64, device=dev)
    y = torch.full((B, max(p, 1)), float("nan"), dtype=torch.float64, device=dev)
    obj = torch.full((B,), float("nan"), dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    it = torch.full((B,), -1, dtype=torch.int32, device=dev)
    ptr = lambda a: C.c_void_p(a.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    common = [0, B, n, m, p, ptr(Pd), ptr(qd), ptr(Gd), ptr(hd), ptr(Ad), ptr(bd), ptr(x), ptr(z), ptr(y), ptr(obj),
              ptr(st), ptr(it)]
    if static:
        rc = lib.mpcqp_solve_dense_static(*common, stream)
    else:
        ni = C.c_long(0)
        nd = lib.mpcqp_dense_workspace(n, m, p, C.byref(ni))
        ws = torch.full((B * nd,), float("nan"), dtype=torch.float64, device=dev)
        iws = torch.full((B * ni.value,), -7, dtype=torch.int32, device=dev)
        rc = lib.mpcqp_solve_dense(*common, ptr(ws), ptr(iws), stream)
    assert rc == _lib.OK, rc
    torch.cuda.synchronize(dev)
    return [a.cpu().numpy() for a in (x, z[:, :m], y[:, :p], obj, st, it)]


def test_static_scratch_matches_caller_workspace_twice():
    from mpc_amd import _lib
    from oracle.qp_oracle import solve_qp
    lib = _lib.load()
    rng = np.random.default_rng(42)
    # first call: 24 QPs of n=40, m=60, p=6; second call, right after it on the same array: 5 smaller QPs
    for (B, n, m, p) in [(24, 40, 60, 6), (5, 18, 12, 3)]:
        data = _problem(rng, B, n, m, p)
        got = _run(lib, True, *data)
        ref = _run(lib, False, *data)
        assert (ref[4] == 0).all() and (got[4] == ref[4]).all() and (got[5] == ref[5]).all()
        # same kernel, same arithmetic order: the two entry points agree to the last bit, far inside the
        # 1e-6 (primal) / 1e-8 (objective) relative bar
        for a, b_ in zip(got[:4], ref[:4]):
            assert np.array_equal(a, b_)
        # and both are the minimiser (oracle on a few of them)
        for i in range(0, B, max(1, B // 3)):
            s = solve_qp(*[d[i] for d in data])
            assert np.abs(got[0][i] - s["x"]).max() <= 1e-6 * max(np.abs(s["x"]).max(), 1e-300)
            assert abs(got[3][i] - s["primal objective"]) <= 1e-8 * max(abs(s["primal objective"]), 1e-300)


def test_static_scratch_rejects_sizes_beyond_its_array():
    import torch
    from mpc_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    ni = C.c_long(0)
    n, m, p = 64, 64, 0
    per = lib.mpcqp_dense_workspace(n, m, p, C.byref(ni))
    too_many = (2 << 20) // per + 1
    d = torch.zeros(8, dtype=torch.float64, device=dev)     # never dereferenced: the size check comes first
    ptr = C.c_void_p(d.data_ptr())
    rc = lib.mpcqp_solve_dense_static(0, too_many, n, m, p, *([ptr] * 12), C.c_void_p(0))
    assert rc == _lib.ENOTIMPL
    assert lib.mpcqp_solve_dense_static(0, 0, n, m, p, *([ptr] * 12), C.c_void_p(0)) == _lib.EINVAL
