import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The form-equivalence tests of this suite (bit-identity of queue / cross-structure launches with single launches, tight form-vs-
# form tolerances) were written for the lane-per-half forms; the row-cooperative form (another elimination order: round-off-level
# differences) has its own tests (tests/test_coop.py, which re-enable its default range on a context of their own).  The library
# itself reads no environment: the Python layer forwards this variable as the context option "coop".
os.environ.setdefault("MTG_COOP", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def host_emu():
    """ctypes handle of the host emulation of the device lane code (test infrastructure)."""
    import ctypes
    so = os.path.join(ROOT, "tests", "libmtg_host_emu.so")
    src = os.path.join(ROOT, "tests", "host_emu.cpp")
    deps = [src] + [os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc", f)
                    for f in ("mtg_lane.h", "mtg_tables.inc", "mtg_variants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    lib.mtg_emu_run.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int, ip]
    lib.mtg_emu_run.restype = ctypes.c_int
    return lib
