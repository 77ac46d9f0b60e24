"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/mpcqp.h
declares (no compute call is made -- there is no GPU here), and the host layer fails loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mpcqp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcqp_[a-z_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from mpc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpcqp.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))


def test_struct_layouts_match_header():
    """ctypes structures carry the same fields, in order, as the C structs."""
    from mpc_amd import _lib
    txt = open(os.path.join(ROOT, "include", "mpcqp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for cname, cls in [("mpcqp_dims", _lib.Dims), ("mpcqp_factor_opts", _lib.FactorOpts),
                       ("mpcqp_solve_opts", _lib.SolveOpts), ("mpcqp_info", _lib.Info)]:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), txt, flags=re.S).group(1)
        fields = re.findall(r"(?:int|double)\s*\*?\s*([a-z_A-Z0-9]+)\s*;", body)
        assert fields == [f[0] for f in cls._fields_], cname


def test_default_opts_and_null_handle_errors():
    from mpc_amd import _lib
    lib = _lib.load()
    o = _lib.SolveOpts()
    lib.mpcqp_default_solve_opts(ctypes.byref(o))
    assert o.max_iter == 0 and o.polish == 1 and o.stages == 3 and abs(o.alpha - 1.6) < 1e-15
    # host-side argument checks that need no device
    assert lib.mpcqp_set_model(None, None, None) == _lib.EINVAL
    assert lib.mpcqp_solve(None, None, 1, None, None, None, None) == _lib.EINVAL
    h = ctypes.c_void_p()
    bad = _lib.Dims(0, 1, 1, 1, 0)
    assert lib.mpcqp_create(ctypes.byref(h), ctypes.byref(bad), 0) == _lib.EINVAL
    assert b"nx" in lib.mpcqp_last_error(None)


def test_error_code_mapping():
    from mpc_amd import _lib

    class L:
        @staticmethod
        def mpcqp_last_error(h):
            return b"boom"
    with pytest.raises(ValueError):
        _lib.check(L, None, _lib.EINVAL)
    with pytest.raises(ArithmeticError):
        _lib.check(L, None, _lib.ENUMERIC)
    with pytest.raises(NotImplementedError):
        _lib.check(L, None, _lib.ENOTIMPL)
    with pytest.raises(RuntimeError):
        _lib.check(L, None, _lib.EHIP)
    _lib.check(L, None, _lib.OK)


def test_planner_refuses_cpu_device():
    from mpc_amd.planner import Planner
    with pytest.raises(RuntimeError):
        Planner(2, 1, 5, device="cpu")


def test_product_never_imports_oracle():
    """The product package must not route through the oracle (or any CPU solver)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "mpc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
