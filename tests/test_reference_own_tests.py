"""The REFERENCE'S OWN test files run against the replacement (tests/ref_tests/README.md): test/test_polynomial_optimization.cpp
and test/test_polynomial.cpp of /root/reference, compiled unmodified where they lie (by __graft_entry__.build(), only where
/root/reference exists; the binaries travel to the GPU box) against (a) the drop-in veneer include/compat/ and (b) the
reference's own class with the solveLinear() body of INTEGRATION.md section 1.  Every assertion of the reference's linear tests
must hold (TOPT:113-174 checkPath 1e-6, :271-306 cost vs numeric integration, :308-400 extrema, :505-564 constraint packing,
:566-606 time allocation, :688-729 time scaling in the trajectory, :731-741 A^-1 1e-10, :743-787 the MATLAB vector 1e-12) over its
ten parameter sets (:790-880); the two tests that drive nlopt are filtered out."""
import os
import re
import subprocess

import pytest

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tests", "bin")
FILTER = "--gtest_filter=-*UnconstrainedNonlinear*:*.TimeScaling/*"


def run(name, args=(), env=None, timeout=1500):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs /root/reference at build time)")
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=timeout, env=e)
    out = r.stdout
    m = re.search(r"\[==========\] (\d+) tests ran, (\d+) filtered out", out)
    assert m, out[-3000:] + r.stderr[-3000:]
    failed = re.findall(r"^\[  FAILED  \] (\S+)", out, flags=re.M)
    tail = "\n".join(l for l in out.splitlines() if "Failure" in l or "FAILED" in l or "Check failed" in l)[-3000:]
    assert r.returncode == 0 and not failed, tail + r.stderr[-2000:]
    return int(m.group(1)), int(m.group(2))


def test_reference_polynomial_tests_on_the_veneer():
    """test/test_polynomial.cpp (Convolution, FindMinMax: 300 random polynomials of up to 13 coefficients on intervals inside
    [-100, 100], computeMinMax against sampling) on the veneer's Polynomial -- host code only, runs without a GPU."""
    ran, _ = run("polynomial_veneer")
    assert ran == 2


@pytest.mark.gpu
@pytest.mark.parametrize("binary,env", [
    ("polynomial_optimization_veneer", {"MTG_COMPAT_SINGLE_CALLS": "host"}),
    ("polynomial_optimization_veneer", {"MTG_COMPAT_SINGLE_CALLS": "device"}),
    ("polynomial_optimization_refclass", {"MTG_REF_TESTS_BACKEND": "host"}),
    ("polynomial_optimization_refclass", {"MTG_REF_TESTS_BACKEND": "device"}),
])
def test_reference_polynomial_optimization_tests(binary, env):
    ran, filtered = run(binary, [FILTER], env)
    # 10 parameter sets x 10 TEST_P patterns, minus the two nlopt-driven patterns
    assert ran == 80 and filtered == 20, (ran, filtered)
