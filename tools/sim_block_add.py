"""CPU experiment (NumPy): how many dependent steps does the dual active-set method of polish_quad.hip need on config 3
when several violated bounds are added per step ("block add": accepted only if every multiplier of the enlarged working
set keeps its sign, else the ordinary single add with its ratio test)?  Prints step statistics per rule; no GPU.
Usage: python tools/sim_block_add.py [instances] [x0 scale]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_amd.configs import make_config          # noqa: E402
from oracle import mpc_oracle as mo              # noqa: E402  (tools/ are experiments, not product)


def gi(Hi, xu, lb, ub, m, frac, maxblk=99):
    n = xu.size
    A, sg = [], []
    lam = np.zeros(0)
    x = xu.copy()
    steps = fails = blocks = 0
    inA = np.zeros(n, bool)
    while True:
        vu, vl = x - ub, lb - x
        v = np.maximum(vu, vl)
        v[inA] = -1.0
        tol = 1e-12 * np.maximum(1.0, np.maximum(np.abs(ub), np.abs(lb)))
        cand = np.where(v > tol)[0]
        if cand.size == 0:
            break
        cand = cand[np.argsort(-v[cand])]
        best = v[cand[0]]
        if m > 1 and cand.size > 1 and blocks < maxblk:
            P = [c for c in cand[:m] if v[c] >= frac * best]
            if len(P) > 1:
                A2 = A + P
                sg2 = np.array(sg + [1.0 if vu[c] > vl[c] else -1.0 for c in P])
                bnd = np.where(sg2 > 0, ub[A2], lb[A2])
                l2 = np.linalg.solve(Hi[np.ix_(A2, A2)], xu[A2] - bnd)
                if np.all(sg2 * l2 >= 0):
                    A, sg, lam = A2, list(sg2), l2
                    inA[P] = True
                    x = xu - Hi[:, A] @ lam
                    steps += 1; blocks += 1
                    continue
                fails += 1
        p = int(cand[0]); up = vu[p] > vl[p]; sp = 1.0 if up else -1.0
        viol = best
        while True:
            steps += 1
            k = len(A)
            h = Hi[p, A]
            r = np.linalg.solve(Hi[np.ix_(A, A)], h) if k else np.zeros(0)
            d2 = Hi[p, p] - h @ r
            t2 = viol / d2
            t1, jb = 1e300, -1
            for a in range(k):
                rr = sp * r[a]
                if sg[a] * rr > 0:
                    t = lam[a] / rr
                    if t < t1:
                        t1, jb = t, a
            full = t2 <= t1
            t = t2 if full else t1
            lam = lam - t * sp * r
            if full:
                A.append(p); sg.append(sp); lam = np.append(lam, 0.0)
                inA[p] = True
                # exact multipliers on the new set
                bnd = np.where(np.array(sg) > 0, ub[A], lb[A])
                lam = np.linalg.solve(Hi[np.ix_(A, A)], xu[A] - bnd)
                x = xu - Hi[:, A] @ lam
                break
            viol -= t * d2
            inA[A[jb]] = False
            del A[jb]; del sg[jb]
            lam = np.delete(lam, jb)
    return steps, fails, len(A)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    prob = make_config(3, batch=B, seed=0)
    H, F, f0 = mo.condense(prob)
    d = 1.0 / np.sqrt(np.diag(H))
    Hs = H * d[:, None] * d[None, :]
    Hi = np.linalg.inv(0.5 * (Hs + Hs.T)); Hi = 0.5 * (Hi + Hi.T)
    lb = np.tile(prob["u_min"], prob["N"]) / d
    ub = np.tile(prob["u_max"], prob["N"]) / d
    Q = (prob["x0"] * scale) @ F.T + f0
    XU = -(Q * d) @ Hi
    rules = [(1, 0.0, 99), (2, 0.0, 99), (3, 0.0, 99), (4, 0.0, 99), (8, 0.0, 99), (99, 0.0, 99), (4, 0.25, 99), (99, 0.25, 99), (99, 0.5, 99),
             (99, 0.0, 1), (99, 0.0, 2)]
    for m, frac, mb in rules:
        st = np.zeros(B, int); fl = np.zeros(B, int); ka = np.zeros(B, int)
        for b in range(B):
            st[b], fl[b], ka[b] = gi(Hi, XU[b], lb, ub, m, frac, mb)
        # a wave of 8 (sorted by nothing here) lasts as long as its slowest instance
        w8 = st.reshape(-1, 8).max(axis=1).mean() if B % 8 == 0 else float("nan")
        print(f"m={m:3d} frac={frac:4.2f} maxblk={mb:2d}: steps mean {st.mean():5.2f} p99 {np.percentile(st, 99):4.0f} max {st.max():3d} | "
              f"failed block tries mean {fl.mean():4.2f} max {fl.max():2d} | chain (steps+fails) max {(st + fl).max():3d} mean {(st + fl).mean():5.2f} | "
              f"active mean {ka.mean():4.1f} | unsorted wave-of-8 mean steps {w8:5.2f}")


if __name__ == "__main__":
    main()
