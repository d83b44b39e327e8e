"""The C++ compat veneer (include/compat/mav_trajectory_generation/*.h): reference-API host code over the C ABI.
CPU: the headers compile with plain g++ (no HIP headers, no Eigen) and link against libmtg_hip.so.
GPU: the test program mirrors the reference's own hot-path tests through the reference's API."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_veneer")
EXE_DEVICE = os.path.join(ROOT, "tests", "cpp", "test_veneer_device")   # single-trajectory calls forced through the GPU


def build_exe(exe=EXE, extra=()):
    src = os.path.join(ROOT, "tests", "cpp", "test_veneer.cpp")
    hdrs = [os.path.join(ROOT, "include", "compat", "mav_trajectory_generation", f)
            for f in os.listdir(os.path.join(ROOT, "include", "compat", "mav_trajectory_generation"))]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(p) for p in hdrs + [src]):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + list(extra) + ["-I" + os.path.join(ROOT, "include", "compat"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                           "-L" + os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc"), "-lmtg_hip", "-pthread",
                           "-Wl,-rpath,$ORIGIN/../../mav_trajectory_generation_amd/csrc"])


def test_veneer_compiles_host_only_and_links():
    build_exe()
    assert os.path.exists(EXE)
    out = subprocess.run(["nm", "-D", "--undefined-only", EXE], capture_output=True, text=True).stdout
    assert "mtg_solve_linear" in out and "hip" not in out.lower().replace("mtg_hip", "")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["host_backend_for_single_calls", "single_calls_on_device"])
def test_veneer_program_on_gpu(which):
    exe = EXE if which == "host_backend_for_single_calls" else EXE_DEVICE
    build_exe(exe, () if exe == EXE else ("-DMTG_COMPAT_SINGLE_CALLS_ON_DEVICE",))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "VENEER TESTS PASSED" in r.stdout
