"""CPU tests of the drop-in boundary: libmtg_hip.so loads and exports every symbol include/mtg_hip.h declares
(no compute calls without a GPU), argument validation that needs no device, and loud failure without one."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    """The drop-in boundary (mtg_hip.h) and the measurement-knob header next to it (mtg_hip_lab.h)."""
    syms = set()
    for name in ("mtg_hip.h", "mtg_hip_lab.h"):
        txt = open(os.path.join(ROOT, "include", name)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(mtg_[a-z_0-9]+)\s*\(", txt))
    return sorted(syms)


def test_the_library_never_reads_the_environment():
    """Round-3 review: 17 getenv knobs lived in the product library.  They are per-context options now
    (mtg_context_set_option, include/mtg_hip_lab.h); only the Python plumbing forwards MTG_* variables."""
    csrc = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp", ".inc")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    from mav_trajectory_generation_amd import _lib
    lib = _lib.load()
    assert lib.mtg_context_set_option(None, b"no_slab", 1) == -1


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("mtg_context_create", "mtg_plan_create", "mtg_solve_linear", "mtg_update_segments_from_free",
              "mtg_context_sync", "mtg_time_last_solve"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    import ctypes
    from mav_trajectory_generation_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert set(declared_symbols()) == set(_lib.EXPORTS.keys())


def test_status_strings_and_no_device_is_loud():
    import torch
    from mav_trajectory_generation_amd import _lib
    lib = _lib.load()
    assert lib.mtg_status_string(0) == b"ok"
    assert b"greater than zero" in lib.mtg_status_string(-2)
    if not torch.cuda.is_available():
        import ctypes
        h = ctypes.c_void_p()
        assert lib.mtg_context_create(0, None, ctypes.byref(h)) == -5  # MTG_ERR_NO_DEVICE, no silent fallback
        import mav_trajectory_generation_amd as m
        with pytest.raises(RuntimeError):
            m.Context(0)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mav_trajectory_generation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "oracle/" not in txt or f.endswith(".py") and "lives in oracle/" in txt, f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/mtg_hip.h compiles as C99 and a C program links against libmtg_hip.so (argument-validation calls only)."""
    import subprocess
    exe = str(tmp_path / "abi_is_c")
    csrc = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_is_c.c"), "-o", exe, "-L" + csrc, "-lmtg_hip",
                           "-Wl,-rpath," + csrc])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "C ABI ok" in r.stdout


def test_multi_entry_points_validate_arguments_without_a_device():
    """mtg_multi_* (mixed requests): null / empty arguments are rejected before any device work."""
    import ctypes
    from mav_trajectory_generation_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    items = (_lib.MultiItem * 1)()
    assert lib.mtg_multi_create(None, 1, items, 0, ctypes.byref(h)) == -1
    assert lib.mtg_multi_solve(None) == -1
    assert lib.mtg_multi_launch_count(None) == 0
    assert lib.mtg_multi_destroy(None) == 0
    assert ctypes.sizeof(_lib.MultiItem) == 8 + 8 + 8 * 8 + 5 * 8     # plan, batch, layout (8 strides), 5 pointers


def test_python_flag_constants_match_the_header():
    """_lib.py's FLAG_* constants are hand-written copies of the header's enum: every MTG_FLAG_* of include/mtg_hip.h must
    exist there with the same bit."""
    from mav_trajectory_generation_amd import _lib
    txt = open(os.path.join(ROOT, "include", "mtg_hip.h")).read()
    flags = dict(re.findall(r"\bMTG_(FLAG_[A-Z_]+)\s*=\s*1u\s*<<\s*(\d+)", txt))
    assert len(flags) >= 10
    for name, bit in flags.items():
        assert getattr(_lib, name) == 1 << int(bit), name
