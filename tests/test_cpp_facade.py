"""The C++ host facade (include/gpmp2mi_planner.hpp) compiles with plain g++ against the C ABI and
links the product library; without a GPU it must fail loudly, with one it must solve."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpmp2_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cpp", "facade_smoke")


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(os.path.join(ROOT, "include", "gpmp2mi_planner.hpp")):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp"), "-o", EXE,
                               "-L", CSRC, "-lgpmp2mi", f"-Wl,-rpath,{CSRC}"])
    return EXE


def _run():
    return subprocess.run([_build()], capture_output=True, text=True, timeout=300)


def test_facade_builds_and_fails_loudly_without_gpu():
    from gpmp2_amd import engine
    r = _run()
    if engine.Engine().device_count() == 0:
        assert r.returncode == 3 and "EXCEPTION" in r.stdout and "no usable HIP device" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_solves_on_gpu():
    r = _run()
    assert r.returncode == 0 and r.stdout.startswith("OK iterations="), r.stdout + r.stderr
