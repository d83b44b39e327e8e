"""CPU tests of the C++ restatement (oracle/cpu_ref.cpp) that serves as second oracle and as bench.py's timed
cpu_baseline: it must agree with the numpy literal restatement (which is pinned on the reference's golden vector)
and its input generator must be bit-exact with the Python one (real libstdc++ std::mt19937 underneath)."""
import os

import numpy as np
import pytest

import helpers
from oracle import cpu_ref
from oracle import oracle_np as onp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_linear_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def case(name):
    pre = name + "/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def test_generator_is_bit_exact_with_libstdcxx():
    """createRandomVertices (vertex.cpp:27-82): Python Mt19937/generate_canonical vs std::mt19937 + uniform_real."""
    pos, times = cpu_ref.generate(6, 8, 3, 1000)
    for b in range(6):
        vs = onp.create_random_vertices(4, 8, [-10.0] * 3, [10.0] * 3, 1000 + b)
        assert np.array_equal(np.stack([v.get_constraint(0) for v in vs]), pos[b])
        assert np.allclose(onp.estimate_segment_times(vs, 3.0, 5.0), times[b], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_cpp_restatement_matches_numpy_restatement(name):
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    masks = [int(m) for m in c["masks"]]
    co, fr, cost, _ = cpu_ref.solve_batch(n, d, masks, c["times"], c["d_fixed"], nthreads=2)
    tol = 1e-6 if (n == 12 or d < n // 2 - 1) else 1e-9     # two float64 evaluations of ill-conditioned formulas
    if n == 12 and d < n // 2 - 1:
        tol = 1e-4
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < tol
    assert np.allclose(cost, c["cost_lit"], rtol=1e-6)
    if name == "two_vertices":
        assert np.abs(co[0, 0, 0] - c["matlab_coeffs"]).max() < 1e-12   # TOPT:777-780


def test_threads_give_identical_results():
    masks, t, f = helpers.reference_batch(40, 8, 10, 3, 99)
    a = cpu_ref.solve_batch(10, 4, masks, t, f, nthreads=1)
    b = cpu_ref.solve_batch(10, 4, masks, t, f, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
