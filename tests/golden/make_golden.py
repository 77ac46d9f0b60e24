#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.  Run in THIS container
(needs /root/reference); the fixtures (*.npz) are committed, this script is the
record of how they were made.  Never runs on the GPU box.

Two families:

ref_<script>.npz   The exact (P, q, G, h, A, b) that a reference planner script
                   passes to ``cvxopt.solvers.qp`` at its first call, captured by
                   executing the script's own assembly code from /root/reference
                   with a *recording* module named ``cvxopt`` on sys.path.  The
                   recorder performs no arithmetic: ``matrix`` keeps the NumPy
                   array, ``solvers.qp`` stores its arguments and stops the
                   script.  (cvxopt itself is absent from this image, see
                   oracle/qp_oracle.py.)  Each file also stores the certified
                   optimum (x, z, y) from oracle.qp_oracle.solve_qp and its KKT
                   residual.
                   Launch3DPlanner.py cannot execute on NumPy >= 1.24
                   (ragged np.array at :33), so its QP comes from the
                   restatement oracle/reference_qps.py::launch3d_qp instead.

mpc_cfg<k>.npz     Synthetic MPC instances of BASELINE.json configs 1-3, 5 at
                   tiny batch, with certified optimal input sequences.

usage: python tests/golden/make_golden.py
"""
import os
import runpy
import sys
import tempfile
import types

import warnings

import numpy as np

warnings.filterwarnings("ignore", category=DeprecationWarning)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


class _Captured(BaseException):
    """Raised by the recorder to stop the script at its first solvers.qp call
    (BaseException so that SE3Planner.py:198's ``except Exception`` does not
    swallow it)."""


def _install_recorder(store):
    cv = types.ModuleType("cvxopt")

    def matrix(x=None, size=None, tc="d"):
        if isinstance(x, list) and len(x) == 0:
            return np.zeros((0, 1))
        return np.array(x, dtype=float)

    sol = types.ModuleType("cvxopt.solvers")
    sol.options = {}

    def qp(P, q, G=None, h=None, A=None, b=None, *a, **k):
        store["args"] = [None if v is None else np.array(v, dtype=float)
                         for v in (P, q, G, h, A, b)]
        store["options"] = dict(sol.options)
        raise _Captured()

    sol.qp = qp
    cv.matrix = matrix
    cv.solvers = sol
    cv.log = np.log
    cv.exp = np.exp
    sys.modules["cvxopt"] = cv
    sys.modules["cvxopt.solvers"] = sol
    # SE3Planner.py:2 imports idlelib.run (unused); give it an empty module so
    # the import does not drag in tkinter on a headless box.
    idl = types.ModuleType("idlelib"); idl.__path__ = []
    idr = types.ModuleType("idlelib.run")
    sys.modules["idlelib"] = idl
    sys.modules["idlelib.run"] = idr


def capture(script):
    store = {}
    _install_recorder(store)
    os.environ["MPLBACKEND"] = "Agg"
    sys.dont_write_bytecode = True
    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    for m in [m for m in sys.modules if m.startswith("src")]:
        del sys.modules[m]
    devnull = open(os.devnull, "w")
    old = sys.stdout
    sys.stdout = devnull
    try:
        # record what the script hands to the reference's own removeNullConstraints (HelperFuncs.py:41-54; the
        # function itself still does the work): the un-pruned rows are a fixture for the device version
        import importlib
        hf = importlib.import_module("src.Planner.HelperFuncs")
        ref_remove = hf.removeNullConstraints

        def recording_remove(J, c, threshold):
            out = ref_remove(J, c, threshold)
            store.setdefault("remove_null", []).append(dict(J=np.array(J, dtype=float), c=np.array(c, dtype=float),
                                                            threshold=float(threshold),
                                                            J_out=np.array(out[0], dtype=float),
                                                            c_out=np.array(out[1], dtype=float)))
            return out
        hf.removeNullConstraints = recording_remove
        runpy.run_path(os.path.join(REF, "src", "Planner", script), run_name="__main__")
    except _Captured:
        pass
    finally:
        sys.stdout = old
        os.chdir(cwd)
        sys.path.remove(REF)
    if "args" not in store:
        raise RuntimeError(f"{script}: solvers.qp was never reached")
    return store


def save_ref(name, P, q, G, h, A, b, source, options=None, extra=None):
    from oracle.qp_oracle import solve_qp
    sol = solve_qp(P, q, G, h, A, b)
    n = q.size
    out = dict(P=P, q=q.reshape(-1),
               G=np.zeros((0, n)) if G is None else G, h=np.zeros(0) if h is None else h.reshape(-1),
               A=np.zeros((0, n)) if A is None else A, b=np.zeros(0) if b is None else b.reshape(-1),
               x=sol["x"], z=sol["z"], y=sol["y"], objective=sol["primal objective"],
               kkt=sol["kkt"]["max"], source=np.array(source))
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **out)
    print(f"ref_{name}: n={n} m={out['h'].size} p={out['b'].size} status={sol['status']} "
          f"it={sol['iterations']} kkt={sol['kkt']['max']:.2e} obj={sol['primal objective']:.12g}")


def make_reference_fixtures():
    from oracle import reference_qps as rq
    for script, name in [("CollocationBasedPlanner.py", "collocation1d"),
                         ("LaunchPlannerQPFormulation.py", "launch1d"),
                         ("SE3Planner.py", "se3_first"),
                         ("CollocationBasedPlanner2D.py", "collocation2d_iter0")]:
        st = capture(script)
        P, q, G, h, A, b = st["args"]
        extra = None
        if st.get("remove_null"):
            # CollocationBasedPlanner2D.py:338: the stacked equality rows BEFORE the reference's removeNullConstraints, its
            # threshold, and what the reference's own function returned for them (== A, b of the captured call)
            rn = st["remove_null"][0]
            assert np.array_equal(rn["J_out"], A) and np.array_equal(rn["c_out"].reshape(-1), b.reshape(-1))
            extra = dict(A_unpruned=rn["J"], b_unpruned=rn["c"].reshape(-1), null_threshold=rn["threshold"])
            print(f"   removeNullConstraints: {rn['J'].shape[0]} rows in, {rn['J_out'].shape[0]} out (reference's own function)")
        save_ref(name, P, q, G, h, A, b, f"captured from src/Planner/{script} (first solvers.qp call)", extra=extra)
        # the restatement in oracle/reference_qps.py must reproduce the capture bit for bit
        fn = getattr(rq, name + "_qp", None)
        if fn is not None:
            for got, want, nm in zip(fn(), (P, q, G, h, A, b), "PqGhAb"):
                if want is None:
                    assert got is None, nm
                else:
                    assert np.array_equal(np.asarray(got, float).reshape(want.shape), want), (name, nm)
            print(f"   restatement oracle/reference_qps.py::{name}_qp == capture (bit-exact)")
    P, q, G, h, A, b = rq.launch3d_qp()
    save_ref("launch3d", P, q, G, h, A, b,
             "restated from src/Planner/Launch3DPlanner.py:14-199 (script itself fails on NumPy>=1.24 at :33)")


def make_remove_null_fixture():
    """null_rows.npz: inputs and outputs of the REFERENCE's own removeNullConstraints (HelperFuncs.py:41-54,
    imported from /root/reference) -- on the equality rows CollocationBasedPlanner2D.py:338 hands to it (nothing is
    removed there) and on the same rows with null rows injected at seeded places (all-zero rows, rows below the
    threshold, null rows with a non-null right-hand side, which make the reference print a warning)."""
    import contextlib
    import importlib
    import io
    sys.path.insert(0, REF)
    try:
        hf = importlib.import_module("src.Planner.HelperFuncs")
    finally:
        sys.path.remove(REF)
    d = np.load(os.path.join(HERE, "ref_collocation2d_iter0.npz"))
    A, b, thr = d["A_unpruned"], d["b_unpruned"].reshape(-1, 1), float(d["null_threshold"])
    rng = np.random.default_rng(7)
    rows, cols = A.shape
    extra = 40
    J = np.vstack([A, np.zeros((extra, cols))]); c = np.vstack([b, np.zeros((extra, 1))])
    J[rows + 10:rows + 20] = rng.uniform(-1, 1, (10, cols)) * 0.9e-10        # below the threshold everywhere
    J[rows + 20:rows + 25, 3] = 1.1e-10                                       # one coefficient just above it: kept
    c[rows + 5:rows + 10] = rng.standard_normal((5, 1))                       # null row, non-null rhs: warned and removed
    c[rows + 15:rows + 18] = 2.0
    perm = rng.permutation(rows + extra)
    J, c = J[perm], c[perm]
    out = {}
    for tag, (Ji, ci) in (("plain", (A, b)), ("injected", (J, c))):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            Jo, co = hf.removeNullConstraints(Ji, ci, thr)
        text = buf.getvalue()
        out.update({f"{tag}_J": Ji, f"{tag}_c": ci, f"{tag}_J_out": Jo, f"{tag}_c_out": co,
                    f"{tag}_warnings": text.count("non zero objective"),
                    f"{tag}_removed": Ji.shape[0] - Jo.shape[0]})
        print(f"null_rows[{tag}]: {Ji.shape[0]} rows in, {Jo.shape[0]} out, {out[f'{tag}_warnings']} warnings")
    np.savez_compressed(os.path.join(HERE, "null_rows.npz"), threshold=thr,
                        source=np.array("src/Planner/HelperFuncs.py::removeNullConstraints run on the rows of "
                                        "CollocationBasedPlanner2D.py:338 (plain) and with seeded null rows (injected)"), **out)


def make_mpc_fixtures():
    from oracle import mpc_oracle as mo
    from oracle.qp_oracle import solve_box_qp
    from mpc_amd.configs import make_config
    for cfg, nb in [(1, 1), (2, 16), (3, 16)]:
        prob = make_config(cfg, batch=nb, seed=0)
        H, F, f0 = mo.condense(prob)
        U = np.zeros((nb, prob["N"] * prob["nu"])); lam = np.zeros_like(U); obj = np.zeros(nb); kkt = np.zeros(nb)
        lb = np.tile(prob["u_min"], prob["N"]); ub = np.tile(prob["u_max"], prob["N"])
        for i in range(nb):
            q = F @ prob["x0"][i] + f0
            s = solve_box_qp(H, q, lb, ub)
            U[i] = s["x"]; lam[i] = s["lam"]; obj[i] = s["primal objective"]; kkt[i] = s["kkt"]["max"]
        np.savez_compressed(os.path.join(HERE, f"mpc_cfg{cfg}.npz"),
                            A=prob["A"], B=prob["B"], Q=prob["Q"], R=prob["R"], P=prob["P"],
                            u_min=prob["u_min"], u_max=prob["u_max"], x0=prob["x0"],
                            N=prob["N"], H=H, F=F, f0=f0, U=U, lam=lam, objective=obj, kkt=kkt)
        print(f"mpc_cfg{cfg}: n={H.shape[0]} batch={nb} max kkt={kkt.max():.2e} "
              f"active min/max={int((lam != 0).sum(1).min())}/{int((lam != 0).sum(1).max())}")


def _install_solving_recorder(log, every):
    """Like the recorder above, but ``solvers.qp`` answers with the oracle's certified optimum so that the script's
    sequential-QP loop keeps running on its own data; every ``every``-th call is logged."""
    from oracle.qp_oracle import solve_qp
    _install_recorder({})
    sol = sys.modules["cvxopt.solvers"]
    count = [0]

    def qp(P, q, G=None, h=None, A=None, b=None, *a, **k):
        args = [None if v is None else np.array(v, dtype=float) for v in (P, q, G, h, A, b)]
        s = solve_qp(*args)
        if count[0] % every == 0:
            log.append(dict(call=count[0], args=args, x=s["x"], z=s["z"], y=s["y"], obj=s["primal objective"],
                            kkt=s["kkt"]["max"], status=s["status"]))
        count[0] += 1
        n = args[1].size
        return {"x": np.asarray(s["x"], float).reshape(n, 1), "status": s["status"], "z": s["z"], "y": s["y"],
                "primal objective": s["primal objective"]}

    sol.qp = qp
    return count


def run_sweep(script, every):
    log = []
    count = _install_solving_recorder(log, every)
    os.environ["MPLBACKEND"] = "Agg"
    sys.dont_write_bytecode = True
    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    for m in [m for m in sys.modules if m.startswith("src")]:
        del sys.modules[m]
    old = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        runpy.run_path(os.path.join(REF, "src", "Planner", script), run_name="__main__")
    finally:
        sys.stdout = old
        os.chdir(cwd)
        sys.path.remove(REF)
    return log, count[0]


def make_sweep_fixtures():
    """The reference's sequential-QP loops end to end: SE3Planner.py's backward sweeps and
    CollocationBasedPlanner2D.py's SQP iterations, each QP built by the script from the answers to the previous ones.
    Stored: a sample of the QPs (grouped by shape) with their certified optima."""
    for script, name, every in [("SE3Planner.py", "se3", 1), ("CollocationBasedPlanner2D.py", "collocation2d", 1)]:
        import time
        t0 = time.time()
        log, total = run_sweep(script, every)
        groups = {}
        for e in log:
            P, q, G, h, A, b = e["args"]
            key = (q.size, 0 if G is None else G.shape[0], 0 if A is None else A.shape[0])
            groups.setdefault(key, []).append(e)
        out = dict(source=np.array(f"src/Planner/{script} run to completion with solvers.qp answered by oracle.qp_oracle.solve_qp; "
                                   f"every {every}th of {total} calls"), total_calls=total, n_groups=len(groups))
        for gi, ((n, m, pe), es) in enumerate(sorted(groups.items())):
            out[f"g{gi}_call"] = np.array([e["call"] for e in es])
            out[f"g{gi}_P"] = np.stack([e["args"][0] for e in es])
            out[f"g{gi}_q"] = np.stack([e["args"][1].reshape(-1) for e in es])
            out[f"g{gi}_G"] = np.stack([e["args"][2] for e in es]) if m else np.zeros((len(es), 0, n))
            out[f"g{gi}_h"] = np.stack([e["args"][3].reshape(-1) for e in es]) if m else np.zeros((len(es), 0))
            out[f"g{gi}_A"] = np.stack([e["args"][4] for e in es]) if pe else np.zeros((len(es), 0, n))
            out[f"g{gi}_b"] = np.stack([e["args"][5].reshape(-1) for e in es]) if pe else np.zeros((len(es), 0))
            out[f"g{gi}_x"] = np.stack([e["x"] for e in es])
            out[f"g{gi}_obj"] = np.array([e["obj"] for e in es])
            out[f"g{gi}_kkt"] = np.array([e["kkt"] for e in es])
            out[f"g{gi}_optimal"] = np.array([e["status"] == "optimal" for e in es])
            print(f"sweep_{name}: group {gi} n={n} m={m} p={pe}: {len(es)} QPs, optimal {int(out[f'g{gi}_optimal'].sum())}, "
                  f"max kkt {out[f'g{gi}_kkt'].max():.2e}")
        np.savez_compressed(os.path.join(HERE, f"sweep_{name}.npz"), **out)
        print(f"sweep_{name}: {total} solvers.qp calls, {len(log)} stored, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["ref", "mpc"]
    if "ref" in which:
        make_reference_fixtures()
    if "ref" in which or "null" in which:
        make_remove_null_fixture()
    if "mpc" in which:
        make_mpc_fixtures()
    if "sweep" in which:
        make_sweep_fixtures()
