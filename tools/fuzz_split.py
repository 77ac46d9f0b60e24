"""Random block-diagonal models with random one-sided / two-sided input bounds: the component-split solve against the
whole-instance solve and the KKT conditions from H itself.  python tools/fuzz_split.py [first_seed] [count]"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_split as T
from oracle import mpc_oracle as mo

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fails = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(2, 5))
    blocks = [(int(rng.integers(2, 5)), 1) for _ in range(nb)]
    N = int(rng.integers(6, 31)); batch = int(rng.choice([64, 700, 9000, 20000]))
    prob = T._block_problem(blocks, N=N, batch=batch, seed=seed, ubound=float(rng.uniform(0.1, 0.5)), x0_amp=float(rng.uniform(1, 5)), stab=0.9)
    umin, umax = prob["u_min"].copy(), prob["u_max"].copy()
    for c in range(nb):
        k = rng.integers(0, 4)
        if k == 1: umin[c] = -np.inf
        elif k == 2: umax[c] = np.inf
    prob["u_min"], prob["u_max"] = umin, umax
    pl = T._planner(prob); U = pl.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    pu = T._planner(prob, split=False); Uu = pu.solve(prob["x0"]).cpu().numpy().reshape(batch, -1)
    H, F, f0 = mo.condense(prob)
    lb, ub = np.tile(umin, N), np.tile(umax, N)
    q = prob["x0"] @ F.T + f0; G = U @ H + q
    sg = np.maximum(np.abs(q).max(axis=1, keepdims=True), 1.0)
    free = (U < ub - 1e-9) & (U > lb + 1e-9)
    bad = int(((U > ub + 1e-12) | (U < lb - 1e-12)).any(1).sum()) + int(((np.abs(G) * free) > 1e-8 * sg).any(1).sum()) \
        + int((~(((G <= 1e-8 * sg) | (U < ub - 1e-9)) & ((G >= -1e-8 * sg) | (U > lb + 1e-9)))).any(1).sum())
    diff = float(np.abs(U - Uu).max() / max(1.0, np.abs(Uu).max()))
    st = int(pl.info.status.max())
    ok = bad == 0 and diff <= 1e-9 and st == 0
    fails += not ok
    print(f"seed {seed}: blocks {blocks} N {N} batch {batch} comps {pl.split_info()['components']} inf {int(np.isinf(umin).sum())}+{int(np.isinf(umax).sum())}: "
          f"bad {bad} status {st} |split - whole| {diff:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES", fails)
sys.exit(1 if fails else 0)
