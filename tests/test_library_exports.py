"""CPU-side check (no GPU needed): the product library loads and exports every symbol that
include/gpmp2mi.h declares; without a GPU its compute entry points fail loudly instead of
falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpmp2_amd", "csrc", "libgpmp2mi.so")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpmp2mi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpmp2mi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    assert os.path.exists(LIB), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    names = _declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/gpmp2mi.h but not exported: {missing}"


def test_no_silent_fallback_without_gpu():
    from gpmp2_amd import engine, generateArm
    eng = engine.Engine()
    if eng.device_count() > 0:
        pytest.skip("a GPU is present; the loud-failure path is only observable without one")
    with pytest.raises(engine.Gpmp2miError) as ei:
        eng.robot(generateArm("WAMArm"))
    assert ei.value.code == 2
    with pytest.raises(engine.Gpmp2miError):
        eng.joint_limit_factor([-1.0], [1.0], [0.1], np.zeros((1, 1)))


def test_pass_driver_spin_is_bounded():
    """the host spin on the device-mapped pass flags gives up after a wall-clock limit (a hung kernel must not
    hang the caller): pointed at a flag that never flips it returns ERR_TIMEOUT with a message; a set flag is read"""
    import time
    lib = ctypes.CDLL(LIB)
    lib.gpmp2mi_last_error.restype = ctypes.c_char_p
    flag, val = ctypes.c_int(-1), ctypes.c_int(-7)
    t0 = time.perf_counter()
    rc = lib.gpmp2mi_debug_wait_flag(ctypes.byref(flag), 60, ctypes.byref(val))
    el = time.perf_counter() - t0
    assert rc == 6 and b"timed out" in lib.gpmp2mi_last_error()
    assert 0.05 <= el < 2.0
    flag.value = 5
    assert lib.gpmp2mi_debug_wait_flag(ctypes.byref(flag), 60, ctypes.byref(val)) == 0 and val.value == 5


def test_public_max_dof_matches_the_library():
    hdr = open(os.path.join(ROOT, "include", "gpmp2mi.h")).read()
    com = open(os.path.join(ROOT, "gpmp2_amd", "csrc", "common.h")).read()
    pub = int(re.search(r"#define GPMP2MI_MAX_DOF (\d+)", hdr).group(1))
    assert re.search(r"constexpr int MAXD = GPMP2MI_MAX_DOF;", com)       # one constant, not two
    assert pub == 18


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gpmp2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "oracle_core" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
