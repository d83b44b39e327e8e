"""CPU tests pinning the oracle (oracle/oracle_np.py) on the reference's own golden vector and property
tests, and on the mpmath ground truth.  References: TOPT = mav_trajectory_generation/test/test_polynomial_optimization.cpp."""
import os

import numpy as np
import pytest

import helpers
from oracle import oracle_mp as omp
from oracle import oracle_np as onp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_linear_golden.npz"))


def case(name):
    pre = name + "/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def case_names():
    return sorted({k.split("/")[0] for k in GOLD.files})


def test_two_vertices_setup_matches_matlab_vector():
    """TOPT:743-787 TwoVerticesSetup -- the reference's only known-answer vector (n_free == 0 branch)."""
    c = case("two_vertices")
    v0, v1 = onp.Vertex(1), onp.Vertex(1)
    v0.make_start_or_end(0.0, 4)
    v1.make_start_or_end(5.0, 4)
    opt = onp.PolynomialOptimization(10, 1)
    opt.setup_from_vertices([v0, v1], [5.0], 4)
    assert opt.n_free == 0
    opt.solve_linear()
    assert np.abs(opt.segments[0, 0] - c["matlab_coeffs"]).max() < 1e-12
    assert np.abs(c["coeffs_lit"][0, 0, 0] - c["matlab_coeffs"]).max() < 1e-12
    assert np.abs(c["coeffs_mp"][0, 0, 0] - c["matlab_coeffs"]).max() < 1e-12


def test_a_matrix_inversion():
    """TOPT:731-741 AMatrixInversion: Schur-complement inverse vs the true inverse for T = 1..60, tol 1e-10
    (the reference compares with Eigen's dense .inverse(); here the exact rational inverse is the arbiter and
    LAPACK's dense inverse -- itself only 3e-10 accurate at T=1 -- is checked at 1e-9)."""
    from fractions import Fraction
    from tools.proto_fast import _gt
    for t in range(1, 61):
        a = onp.setup_mapping_matrix(10, float(t))
        ai = onp.invert_mapping_matrix(a)
        exact = np.array([[float(x) for x in r] for r in _gt.mat_inv([[Fraction(int(v)) for v in row] for row in a])])
        assert np.abs(ai - exact).max() < 1e-10, t
        assert np.abs(ai - np.linalg.inv(a)).max() < 1e-9, t


def test_base_coefficients_table():
    """polynomial.cpp:145-160: base(n, i) = i! / (i - n)!."""
    import math
    bc = onp.BASE_COEFFICIENTS
    for n in range(6):
        for i in range(12):
            want = math.factorial(i) // math.factorial(i - n) if i >= n else 0
            assert bc[n, i] == want


def test_mt19937_matches_libstdcxx_known_value():
    g = onp.Mt19937(5489)
    for _ in range(9999):
        g.next_u32()
    assert g.next_u32() == 4123659995  # C++ standard [rand.predef]


def test_vertex_generation():
    """TOPT:250-269 VertexGeneration: bounds and constraint counts."""
    for seed in range(5):
        vs = onp.create_random_vertices(4, 10, [-10.0] * 3, [10.0] * 3, seed)
        assert len(vs) == 11
        for i, v in enumerate(vs):
            want = 5 if i in (0, 10) else 1
            assert len(v.constraints) == want
            p = v.get_constraint(0)
            assert np.all(p >= -10) and np.all(p <= 10)
        for a, b in zip(vs[:-1], vs[1:]):
            assert np.linalg.norm(a.get_constraint(0) - b.get_constraint(0)) > 0.2


@pytest.mark.parametrize("dim,k,seed", [(1, 1, 100), (1, 10, 102), (3, 1, 104), (3, 10, 105), (3, 50, 106)])
def test_check_path_and_cost(dim, k, seed):
    """TOPT:271-306: fixed constraints met, C^0..C^4 continuity (1e-6), cost = numeric integral (10 %)."""
    vs = onp.create_random_vertices(4, k, [-10.0] * dim, [10.0] * dim, seed)
    t = onp.estimate_segment_times(vs, 3.0, 5.0)
    opt = onp.PolynomialOptimization(10, dim)
    opt.setup_from_vertices(vs, t, 4)
    opt.solve_linear()
    masks = opt.fixed_mask()
    d_fixed = np.stack(opt.fixed_constraints_compact)[None]
    worst = helpers.check_path(masks, np.array([t]), d_fixed, opt.segments[None])
    assert worst < 1e-6
    # numeric cost: integral of squared snap by Gauss-Legendre per segment
    xs, ws = np.polynomial.legendre.leggauss(20)
    num = 0.0
    for i in range(k):
        tt = 0.5 * t[i] * (xs + 1)
        for d in range(dim):
            s = helpers.evaluate(opt.segments[i, d], tt, 4)
            num += 0.5 * t[i] * np.sum(ws * s * s)
    assert abs(opt.compute_cost() - num) <= 0.1 * num + 1e-12


def test_constraint_packing():
    """TOPT:505-564 ConstraintPacking: [d_F; d_P] -> p = A^-1 M d -> A p -> M^+ round trip, 1e-6."""
    for i in range(5):
        vs = onp.create_random_vertices(4, 10, [-50.0] * 3, [50.0] * 3, 12345 + i)
        t = onp.estimate_segment_times(vs, 3.0, 5.0)
        opt = onp.PolynomialOptimization(10, 3)
        opt.setup_from_vertices(vs, t)
        opt.solve_linear()
        m, a_inv, a, m_pinv = opt.get_m(), opt.get_a_inverse(), opt.get_a(), opt.get_m_pinv()
        for d in range(3):
            d_all = np.concatenate([opt.fixed_constraints_compact[d], opt.free_constraints_compact[d]])
            p = a_inv @ m @ d_all
            d_re = m_pinv @ (a @ p)
            assert np.abs(d_all - d_re).max() < 1e-6
            for j in range(10):
                assert np.abs(opt.segments[j, d] - p[j * 10:(j + 1) * 10]).max() < 1e-6


def test_kkt_optimality():
    """Not asserted by the reference (SURVEY.md section 4 gap): R_PP d_P + R_PF d_F = 0."""
    vs = onp.create_random_vertices(4, 8, [-10.0] * 3, [10.0] * 3, 7)
    t = onp.estimate_segment_times(vs, 3.0, 5.0)
    opt = onp.PolynomialOptimization(10, 3)
    opt.setup_from_vertices(vs, t, 4)
    opt.solve_linear()
    r = opt.construct_r()
    nf = opt.n_fixed
    for d in range(3):
        res = r[nf:, nf:] @ opt.free_constraints_compact[d] + r[nf:, :nf] @ opt.fixed_constraints_compact[d]
        scale = np.abs(r[nf:, :nf]).max() * np.abs(opt.fixed_constraints_compact[d]).max()
        assert np.abs(res).max() < 1e-9 * scale


@pytest.mark.parametrize("name", [n for n in case_names()])
def test_oracle_reproduces_golden_fixture(name):
    """The committed fixtures are regenerated bit-for-bit by the oracle (guards against silent oracle drift)."""
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    co, fr, j = onp.solve_batch(n, d, list(c["masks"]), c["times"], c["d_fixed"])
    assert np.array_equal(co, c["coeffs_lit"]) or helpers.poly_relerr(co, c["coeffs_lit"]) < 1e-13


@pytest.mark.parametrize("name,tol", [("readme", 1e-9), ("config2", 1e-9), ("config5", 1e-9), ("two_vertices", 1e-12),
                                      ("topt_D3_d4_K10_s105", 1e-9), ("config4_N8_K4", 1e-9),
                                      ("config4_N12_K8", 5e-7), ("feas_pos_N12", 5e-7), ("feas_yaw_N12", 1e-5),
                                      ("topt_D3_d2_K5_s109", 5e-7), ("topt_D3_d3_K5_s110", 1e-8)])
def test_literal_oracle_vs_mpmath_truth(name, tol):
    """float64 evaluation of the reference's formulas vs the 50-digit solve: ~1e-11 norm-wise for N=10/snap,
    ~1e-8 for N=12 or d < h-1 (cond(A) 1e11..1e17) -- documents what '1e-9 vs Eigen' can mean."""
    c = case(name)
    assert helpers.poly_relerr(c["coeffs_lit"], c["coeffs_mp"]) < tol


def test_readme_example_values():
    """SURVEY.md 8(c) config-1 vector (regenerated here, mpmath): nfabian times and segment-0 x coefficients."""
    c = case("readme")
    assert np.allclose(c["times"][0], [3.970847833173347, 3.8241301415334297], rtol=0, atol=1e-14)
    want = [0, 0, 0, 0, 0, 1.339252819662407e-02, -7.845546057749868e-03, 1.954568943684918e-03,
            -2.392908039915994e-04, 1.181394415296818e-05]
    assert np.abs(c["coeffs_mp"][0, 0, 0] - want).max() < 1e-15


def test_mp_single_matches_batch():
    c = case("readme")
    co, _, _ = omp.solve(10, 4, list(c["masks"]), c["times"][0], c["d_fixed"][0])
    assert np.array_equal(co, c["coeffs_mp"][0])


def test_sampling_restatement_vs_numpy_polynomials():
    """oracle sample_batch (Trajectory::evaluate + Polynomial::evaluate restatement) vs numpy's own polynomial
    derivative evaluation on the README example."""
    c = case("readme")
    coeffs, times = c["coeffs_mp"], c["times"]
    out, nv = onp.sample_batch(coeffs, times, 0.0, 0.37, 25, 5)
    total = times[0].sum()
    assert nv[0] == int(total / 0.37) + 1
    for s in range(25):
        t = min(0.37 * s, total)
        seg = 0 if t < times[0, 0] else 1
        local = t - (times[0, 0] if seg else 0.0)
        for der in range(5):
            for d in range(3):
                p = np.polynomial.Polynomial(coeffs[0, seg, d]).deriv(der) if der else np.polynomial.Polynomial(coeffs[0, seg, d])
                assert abs(out[0, s, der, d] - p(local)) <= 1e-12 * (1 + abs(p(local)))
