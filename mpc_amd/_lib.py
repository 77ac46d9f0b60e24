"""ctypes binding of libmpcqp.so (include/mpcqp.h).  There is no fallback: if the
HIP library is missing or a symbol is absent the import of the product path
fails loudly (RuntimeError), never silently on the CPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpcqp.so")

OK, EINVAL, ENUMERIC, EHIP, ESTATE, ENOTIMPL = range(6)
SOLVED, SOLVED_INACCURATE, MAX_ITER = 0, 1, 2
STATUS_NAMES = {SOLVED: "solved", SOLVED_INACCURATE: "solved inaccurate", MAX_ITER: "maximum iterations reached"}


class Dims(C.Structure):
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int),
                ("n_models", C.c_int), ("time_varying", C.c_int)]


class FactorOpts(C.Structure):
    _fields_ = [("rho", C.c_double), ("no_sync", C.c_int), ("keep_state", C.c_int)]


class SolveOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("min_iter", C.c_int), ("check_every", C.c_int),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("alpha", C.c_double),
                ("polish", C.c_int), ("polish_refine", C.c_int), ("polish_max_steps", C.c_int),
                ("warm_start", C.c_int), ("stages", C.c_int), ("keep_state", C.c_int)]


class DenseOpts(C.Structure):
    _fields_ = [("max_steps", C.c_int), ("prox_iters", C.c_int), ("refine", C.c_int), ("sigma_rel", C.c_double)]


class Info(C.Structure):
    _fields_ = [("iters", C.c_void_p), ("status", C.c_void_p), ("obj", C.c_void_p),
                ("prim_res", C.c_void_p), ("dual_res", C.c_void_p), ("polish_steps", C.c_void_p),
                ("lam", C.c_void_p)]


# every symbol include/mpcqp.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mpcqp_create", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Dims), C.c_int]),
    ("mpcqp_destroy", C.c_int, [C.c_void_p]),
    ("mpcqp_last_error", C.c_char_p, [C.c_void_p]),
    ("mpcqp_default_solve_opts", None, [C.POINTER(SolveOpts)]),
    ("mpcqp_set_model", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mpcqp_set_cost", C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    ("mpcqp_set_constraints", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("mpcqp_set_strides", C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_long]),
    ("mpcqp_set_state_constraints", C.c_int, [C.c_void_p] * 6 + [C.c_int]),
    ("mpcqp_set_general_opts", C.c_int, [C.c_void_p, C.c_int, C.c_long]),
    ("mpcqp_general_rows", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("mpcqp_set_split", C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    ("mpcqp_adapt_rho", C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]),
    ("mpcqp_pin", C.c_int, [C.c_void_p, C.c_int]),
    ("mpcqp_split_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    ("mpcqp_assemble", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mpcqp_factor", C.c_int, [C.c_void_p, C.POINTER(FactorOpts), C.c_void_p]),
    ("mpcqp_solve", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Info),
                              C.POINTER(SolveOpts), C.c_void_p]),
    ("mpcqp_shift", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mpcqp_predict", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("mpcqp_recenter", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mpcqp_shift_recenter", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mpcqp_get", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p]),
    ("mpcqp_dense_workspace", C.c_long, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]),
    ("mpcqp_remove_null_rows", C.c_int, [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_double] + [C.c_void_p] * 5),
    ("mpcqp_ltv_bicycle", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 13),
    ("mpcqp_ltv_bicycle_apply", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 7),
    ("mpcqp_ltv_bicycle_step", C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 4),
    ("mpcqp_solve_dense_strided", C.c_int, [C.c_int] * 5 + [C.c_void_p, C.c_long] * 6 + [C.c_void_p] * 8 + [C.c_void_p]),
    ("mpcqp_ltv_bicycle_advance", C.c_int, [C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 7),
    ("mpcqp_default_dense_opts", None, [C.POINTER(DenseOpts)]),
    ("mpcqp_solve_dense_opts", C.c_int, [C.c_int] * 5 + [C.c_void_p, C.c_long] * 6 + [C.c_void_p] * 8 +
     [C.POINTER(DenseOpts), C.c_void_p]),
    ("mpcqp_dense_shared_setup", C.c_int, [C.c_int] * 4 + [C.c_void_p] * 6),
    ("mpcqp_dense_shared_solve", C.c_int, [C.c_int] * 5 + [C.c_void_p] * 12),
    ("mpcqp_solve_dense", C.c_int, [C.c_int] * 5 + [C.c_void_p] * 14 + [C.c_void_p]),
    ("mpcqp_solve_dense_static", C.c_int, [C.c_int] * 5 + [C.c_void_p] * 12 + [C.c_void_p]),
    ("mpcqp_rows_workspace", C.c_long, [C.c_int, C.c_int, C.POINTER(C.c_long)]),
    ("mpcqp_rows_setup", C.c_int, [C.c_int] * 3 + [C.c_void_p] * 5),
    ("mpcqp_rows_solve", C.c_int, [C.c_int] * 4 + [C.c_void_p] * 10),
    ("mpcqp_rows_solve_mpc_warm", C.c_int, [C.c_int] * 5 + [C.c_void_p] * 13),
    ("mpcqp_rows_solve_mpc", C.c_int, [C.c_int] * 5 + [C.c_void_p] * 13),
]

_lib = None


def load():
    """Load libmpcqp.so and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mpc_amd/csrc`).  mpc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libmpcqp.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, handle, rc):
    """Map a non-zero return code to the exception class the reference's callers
    expect (SE3Planner.py:107,198 catches Exception subclasses)."""
    if rc == OK:
        return
    msg = lib.mpcqp_last_error(handle)
    msg = msg.decode() if msg else f"mpcqp error {rc}"
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENUMERIC:
        raise ArithmeticError(msg)
    if rc == ENOTIMPL:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
