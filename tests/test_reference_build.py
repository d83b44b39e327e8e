"""Pins the oracles on the REFERENCE ITSELF: oracle/_ref/libmtg_ref.so is the reference's own
PolynomialOptimization<N> / Vertex / Segment / Trajectory code compiled from /root/reference where it lies
(oracle/Makefile `ref`, Eigen/glog replaced by the container stand-ins of oracle/ref_shim -- see oracle/ref_linear.py).

Two layers:
  * fixture tests -- tests/golden/reference_solve_linear.npz (outputs of that library, generated here by
    tests/golden/make_reference_golden.py) against the restatements; run everywhere, library or not;
  * live tests    -- call the library directly (skipped only where oracle/_ref is absent: a checkout that has neither
    /root/reference nor the prebuilt file).
Tolerances: the restatements and the compiled reference evaluate the same float64 formulas with different (all
backward-stable) inverse / QR kernels, so they differ by round-off x cond: <= 1e-9 norm-wise for N <= 10 with
d = h-1 (measured <= 7e-11), looser for N = 12 / d < h-1 exactly as for the mpmath comparison in test_oracle.py.
"""
import os

import numpy as np
import pytest

import helpers
from oracle import cpu_ref
from oracle import oracle_extrema as ox
from oracle import oracle_np as onp
from oracle import ref_linear

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "solve_linear_golden.npz"))
REF = np.load(os.path.join(HERE, "golden", "reference_solve_linear.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})
live = pytest.mark.skipif(not ref_linear.available(), reason="oracle/_ref/libmtg_ref.so not built (needs /root/reference)")


def tol_for(n, d):
    if n == 12 and d < n // 2 - 1:
        return 1e-5
    if n == 12 or d < n // 2 - 1:
        return 5e-8
    return 1e-9


def inputs(name):
    return (int(GOLD[f"{name}/n"]), int(GOLD[f"{name}/d"]), [int(m) for m in GOLD[f"{name}/masks"]],
            GOLD[f"{name}/times"], GOLD[f"{name}/d_fixed"])


# ------------------------------------------------------------------------------------------------ fixture layer
def test_fixture_covers_every_golden_case():
    assert sorted({k.split("/")[0] for k in REF.files}) == NAMES


def test_reference_reproduces_its_own_known_answer_vector():
    """TOPT:777-780 (TwoVerticesSetup, MATLAB coefficients) through the compiled reference."""
    assert np.abs(REF["two_vertices/coeffs_ref"][0, 0, 0] - GOLD["two_vertices/matlab_coeffs"]).max() < 1e-12


@pytest.mark.parametrize("name", NAMES)
def test_numpy_restatement_vs_reference_outputs(name):
    n, d, masks, times, d_fixed = inputs(name)
    ref_c, ref_f, ref_j = REF[f"{name}/coeffs_ref"], REF[f"{name}/d_free_ref"], REF[f"{name}/cost_ref"]
    lit_c, lit_f, lit_j = GOLD[f"{name}/coeffs_lit"], GOLD[f"{name}/d_free_lit"], GOLD[f"{name}/cost_lit"]
    tol = tol_for(n, d)
    assert helpers.poly_relerr(lit_c, ref_c) < tol
    if ref_f.size and d == n // 2 - 1:   # d < h-1: high free derivatives are barely determined (cond(R_PP) ~ 1e8)
        assert np.abs(lit_f - ref_f).max() <= 10 * tol * max(1.0, np.abs(ref_f).max())
    assert np.allclose(lit_j, ref_j, rtol=max(1e-8, tol))
    # the reference's outputs satisfy its own checkPath property (TOPT:113-174) on these inputs
    assert helpers.check_path(masks, times, d_fixed, ref_c) < 1e-6


@pytest.mark.parametrize("name", [n for n in NAMES if f"{n}/coeffs_mp" in GOLD.files])
def test_reference_outputs_vs_mpmath_truth(name):
    """The compiled reference is as far from the 50-digit solve as the restatement is (same float64 formulas)."""
    n, d = int(GOLD[f"{name}/n"]), int(GOLD[f"{name}/d"])
    e_ref = helpers.poly_relerr(REF[f"{name}/coeffs_ref"], GOLD[f"{name}/coeffs_mp"])
    e_lit = helpers.poly_relerr(GOLD[f"{name}/coeffs_lit"], GOLD[f"{name}/coeffs_mp"])
    assert e_ref < tol_for(n, d)
    assert e_ref < 30 * max(e_lit, 1e-14) + 1e-12


@pytest.mark.parametrize("name", ["config2", "config5", "config4_N8_K4", "readme", "topt_D3_d4_K10_s105"])
def test_cpp_restatement_vs_reference_outputs(name):
    n, d, masks, times, d_fixed = inputs(name)
    co, fr, cost, _ = cpu_ref.solve_batch(n, d, masks, times, d_fixed)
    assert helpers.poly_relerr(co, REF[f"{name}/coeffs_ref"]) < tol_for(n, d)
    assert np.allclose(cost, REF[f"{name}/cost_ref"], rtol=1e-8)


@pytest.mark.parametrize("name", ["config2", "config5", "config4_N12_K8", "feas_yaw_N12", "topt_D1_d4_K10_s102"])
def test_sampling_restatement_vs_reference_evaluate(name):
    """oracle_np.trajectory_evaluate vs the reference's Trajectory::evaluate on the reference's own coefficients."""
    n, d, masks, times, _ = inputs(name)
    seg, t = REF[f"{name}/coeffs_ref"][0], times[0]
    for der, key in ((0, "sample_pos"), (2, "sample_acc")):
        want = REF[f"{name}/{key}"]
        got = np.array([onp.trajectory_evaluate(seg, t, x, der) for x in REF[f"{name}/sample_t"]])
        assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("name", ["config2", "config5", "config4_N8_K4", "readme"])
def test_extrema_restatement_vs_reference_minmax(name):
    """oracle_extrema (restatement + reference rpoly back end) vs Trajectory::computeMinMaxMagnitude itself."""
    n, d, masks, times, _ = inputs(name)
    seg, t = REF[f"{name}/coeffs_ref"][0], times[0]
    backend = "ref" if ox.ref_available() else "numpy"
    for der in (1, 2):
        mn, mx, per = ox.trajectory_min_max_magnitude(seg, t, der, None, backend)
        want, want_per = REF[f"{name}/minmax_d{der}"], REF[f"{name}/minmax_per_segment_d{der}"]
        scale = abs(want[4])
        tol = 1e-12 if backend == "ref" else 1e-8
        assert abs(mx[1] - want[4]) <= tol * scale and abs(mn[1] - want[1]) <= max(tol, 1e-9) * scale
        assert np.abs(np.asarray(per)[:, 3] - want_per[:, 3]).max() <= tol * scale
        if backend == "ref":
            assert mx[2] == int(want[5]) and abs(mx[0] - want[3]) <= 1e-9


# ------------------------------------------------------------------------------------------------ live layer
@live
@pytest.mark.parametrize("name", NAMES)
def test_committed_fixture_is_what_the_library_produces(name):
    n, d, masks, times, d_fixed = inputs(name)
    co, fr, cost, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    assert helpers.poly_relerr(co, REF[f"{name}/coeffs_ref"]) < 1e-13
    assert np.allclose(cost, REF[f"{name}/cost_ref"], rtol=1e-12)


@live
@pytest.mark.parametrize("n,d", [(10, 4), (8, 3), (12, 5), (10, 2), (6, 2), (4, 1), (2, 0)])
def test_segment_matrix_builders(n, d):
    """setupMappingMatrix / computeQuadraticCostJacobian: same formulas, same libm -> equal to the last bit or ulp;
    invertMappingMatrix: Schur complement with a differently pivoted h x h inverse -> cond(D) x eps."""
    for t in (0.37, 1.0, 2.5, 7.75):
        a, ai, q = ref_linear.segment_matrices(n, d, t)
        assert np.array_equal(a, onp.setup_mapping_matrix(n, t))
        q_lit = onp.compute_quadratic_cost_jacobian(n, d, t)
        assert np.abs(q - q_lit).max() <= 4e-16 * np.abs(q_lit).max()
        ai_lit = onp.invert_mapping_matrix(onp.setup_mapping_matrix(n, t))
        assert np.abs(ai - ai_lit).max() <= 1e-9 * np.abs(ai_lit).max()
        assert np.abs(a @ ai - np.eye(n)).max() < 1e-6


@live
@pytest.mark.parametrize("n,d,k,dim,masks", [(10, 4, 8, 3, None), (10, 4, 5, 2, [31, 7, 1, 3, 5, 31]),
                                             (12, 5, 4, 3, [31, 1, 1, 1, 31]), (8, 3, 6, 1, None)])
def test_reordering_matrix_and_r(n, d, k, dim, masks):
    """M (LIN:182-260) is pure indexing -> identical; R = M^T H M (LIN:308-336) to round-off of the H products."""
    masks, times, d_fixed = helpers.reference_batch(1, k, n, dim, 99, masks)
    m_ref, r_ref, nf, npf = ref_linear.m_and_r(n, d, masks, times[0], d_fixed[0])
    verts = [onp.Vertex(dim) for _ in range(k + 1)]
    col = 0
    for v in range(k + 1):
        for p in range(n // 2):
            if (masks[v] >> p) & 1:
                verts[v].add_constraint(p, d_fixed[0, :, col])
                col += 1
    opt = onp.PolynomialOptimization(n, dim)
    opt.setup_from_vertices(verts, times[0], d)
    assert (nf, npf) == (opt.n_fixed, opt.n_free)
    assert np.array_equal(m_ref[:, :nf + npf], opt.get_m())
    r_lit = opt.construct_r()
    assert np.abs(r_ref[:nf + npf, :nf + npf] - r_lit).max() <= 1e-7 * np.abs(r_lit).max()
    assert opt.fixed_mask() == [m & ((1 << (n // 2)) - 1) for m in masks]


@live
@pytest.mark.parametrize("k,dim,seed", [(8, 3, 0), (8, 3, 12345), (1, 1, 7), (50, 4, 2**31 + 5)])
def test_generators_bit_exact(k, dim, seed):
    """createRandomVertices + estimateSegmentTimesNfabian: the three implementations (reference, numpy restatement
    with its own mt19937, C++ port through libstdc++) agree to the last bit."""
    pos, times = ref_linear.random_vertices(4, k, dim, -10.0, 10.0, seed)
    vs = onp.create_random_vertices(4, k, [-10.0] * dim, [10.0] * dim, seed)
    assert np.array_equal(pos, np.array([v.get_constraint(0) for v in vs]))
    # positions bit for bit; times through numpy's norm (different summation order for dim >= 4): last-bit level
    assert np.allclose(times, np.array(onp.estimate_segment_times_nfabian(vs, 3.0, 5.0)), rtol=5e-16, atol=0)
    pos2, times2 = cpu_ref.generate(1, k, dim, seed)
    assert np.array_equal(pos2[0], pos) and np.array_equal(times2[0], times)


@live
@pytest.mark.parametrize("n,d,k,dim,masks,bsz", [
    (10, 4, 8, 3, None, 200),                                # BASELINE config 2/3 shape
    (10, 4, 16, 4, [31] + [7] * 15 + [31], 40),              # config 5 shape (pos+vel+acc interior)
    (8, 3, 4, 3, None, 60), (8, 3, 32, 3, None, 10),         # config 4 corners
    (10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31], 30),              # ragged per-vertex masks
    (10, 4, 1, 3, None, 30),                                 # n_free == 0 (LIN:343-349)
    (10, 4, 3, 2, [3, 1, 1, 7], 30),                         # free end-vertex slots
    (6, 2, 4, 3, None, 30), (4, 1, 3, 2, None, 30), (2, 0, 3, 2, None, 30),
    # chain lengths between the BASELINE ones (odd K included): every K = 2 .. 32 has its own GPU kernel since round 2
    (10, 4, 5, 3, None, 30), (10, 4, 11, 3, None, 20), (10, 4, 20, 3, None, 12), (10, 4, 27, 3, None, 12),
    (8, 3, 7, 3, None, 30), (8, 3, 21, 3, None, 12), (10, 4, 50, 3, None, 12),
])
def test_fresh_batches_restatements_vs_live_reference(n, d, k, dim, masks, bsz):
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 20260924, masks)
    ref_c, ref_f, ref_j, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    lit_c, lit_f, lit_j = onp.solve_batch(n, d, masks, times[:12], d_fixed[:12])
    cpp_c, cpp_f, cpp_j, _ = cpu_ref.solve_batch(n, d, masks, times, d_fixed)
    # free slots on the end vertices (the N = 12 / yaw patterns of test_feasibility.cpp) condition R_PP worse
    tol = 1e-9 if (masks[0] == masks[-1] == (1 << (n // 2)) - 1) else 1e-8
    assert helpers.poly_relerr(lit_c, ref_c[:12]) < tol
    assert helpers.poly_relerr(cpp_c, ref_c) < tol
    assert np.allclose(cpp_j, ref_j, rtol=1e-8) and np.allclose(lit_j, ref_j[:12], rtol=1e-8)
    if ref_f.size:
        assert np.abs(cpp_f - ref_f).max() <= 1e-8 * max(1.0, np.abs(ref_f).max())


@live
def test_set_free_constraints_path():
    """setFreeConstraints (LIN:500-508): feeding the solved d_P back reproduces the segments; a perturbed d_P gives
    the restatement's coefficients and a larger cost."""
    masks, times, d_fixed = helpers.reference_batch(10, 8, 10, 3, 555)
    co, fr, cost, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed)
    co2, _, cost2, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed, d_free_in=fr)
    assert helpers.poly_relerr(co2, co) < 1e-14 and np.allclose(cost2, cost, rtol=1e-13)
    co3, _, cost3, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed, d_free_in=fr * 1.02)
    assert np.all(cost3 > cost)
    for b in range(3):
        verts = [onp.Vertex(3) for _ in range(9)]
        col = 0
        for v in range(9):
            for p in range(5):
                if (masks[v] >> p) & 1:
                    verts[v].add_constraint(p, d_fixed[b, :, col])
                    col += 1
        opt = onp.PolynomialOptimization(10, 3)
        opt.setup_from_vertices(verts, times[b], 4)
        opt.set_free_constraints([fr[b, dd] * 1.02 for dd in range(3)])
        assert helpers.poly_relerr(opt.segments[None], co3[b][None]) < 1e-9


@live
def test_evaluate_range_semantics():
    """Trajectory::evaluateRange (trajectory.cpp:81-141): accumulated sample times, sample count, segment hand-over --
    against the closed-form grid of oracle_np.sample_batch where both define a sample."""
    masks, times, d_fixed = helpers.reference_batch(3, 5, 10, 3, 808)
    co, _, _, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed)
    for b in range(3):
        total = float(times[b].sum())
        dt = total / 64.0
        vals, st = ref_linear.evaluate_range(co[b], times[b], 0.0, total, dt, 1)
        assert abs(len(st) - 65) <= 1 and st[0] == 0.0
        want, n_valid = onp.sample_batch(co[b][None], times[b][None], 0.0, dt, len(st), 2)
        # the reference accumulates t += dt (drift ~ 1e-15 * count); compare through the local slope
        scale = np.abs(want[0, :, 1]).max()
        assert np.abs(vals[:60] - want[0, :60, 1]).max() <= 1e-10 * scale


@live
def test_scale_segment_times_restatement_vs_reference():
    masks, times, d_fixed = helpers.reference_batch(12, 6, 10, 3, 4242)
    co, _, _, _ = ref_linear.solve_batch(10, 4, masks, times, d_fixed)
    backend = "ref" if ox.ref_available() else "numpy"
    n_scaled = 0
    for b in range(12):
        c_ref, t_ref, ok_ref = ref_linear.scale_segment_times_to_meet_constraints(co[b], times[b], 2.0, 2.5)
        ok, segs, new_t, ns = ox.scale_segment_times_to_meet_constraints(co[b], times[b], 2.0, 2.5, backend)
        n_scaled += ns
        assert ok == ok_ref
        assert np.max(np.abs(new_t - t_ref) / t_ref) <= 1e-10
        cs = np.abs(c_ref).max(axis=-1, keepdims=True)
        assert np.max(np.abs(segs - c_ref) / cs) <= 1e-10
    assert n_scaled > 0


# ------------------------------------------------------------------------------------------------ real-Eigen layer
@pytest.mark.skipif(not ref_linear.eigen_build_available(),
                    reason="oracle/_ref/libmtg_ref_eigen.so absent: this image has no Eigen (make -C oracle ref_eigen EIGEN_DIR=...)")
@pytest.mark.parametrize("name", NAMES)
def test_real_eigen_build_vs_stand_in(name):
    """the anchor's last bits are Eigen's (inverse(), sparse products, SparseQR + COLAMD), which the container
    stand-ins only follow by algorithm class.  On a box that has Eigen, `make -C oracle ref_eigen EIGEN_DIR=...` builds the SAME
    reference sources against it; this test then pins the committed stand-in outputs to the real-Eigen outputs within the
    reference's own distance to the 50-digit truth (the tolerance every comparison against the reference uses), and checks
    the known-answer vector through real Eigen."""
    n, d, masks, times, d_fixed = inputs(name)
    ref_linear.use_eigen_build(True)
    try:
        co, fr, cost, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    finally:
        ref_linear.use_eigen_build(False)
    tol = tol_for(n, d)
    assert helpers.poly_relerr(co, REF[f"{name}/coeffs_ref"]) < tol
    assert np.allclose(cost, REF[f"{name}/cost_ref"], rtol=1e-8)
    if name == "two_vertices":
        assert np.abs(co[0, 0, 0] - GOLD["two_vertices/matlab_coeffs"]).max() < 1e-12


# ------------------------------------------------------------------------------------------------ row N2: the Mellinger step
MEL = np.load(os.path.join(HERE, "golden", "reference_mellinger.npz"))
MEL_NAMES = sorted({k.split("/")[0] for k in MEL.files})
live_nl = pytest.mark.skipif(not ref_linear.nonlinear_available(), reason="oracle/_ref/libmtg_ref_nl.so not built (needs /root/reference)")


def mel_inputs(name):
    return (int(MEL[f"{name}/n"]), int(MEL[f"{name}/d"]), [int(m) for m in MEL[f"{name}/masks"]], MEL[f"{name}/times"],
            MEL[f"{name}/d_fixed"])


@pytest.mark.parametrize("name", MEL_NAMES)
def test_mellinger_restatement_vs_reference_member(name):
    """oracle_np.mellinger_cost_gradient against the outputs of the reference's OWN getCostAndGradientMellinger (NL:287-364) run
    in the build container (tests/golden/make_reference_mellinger_golden.py): cost 1e-9, gradient 1e-9 of its scale -- the two
    evaluate the same float64 formulas; a forward difference of costs that agree to round-off x cond divided by h = 0.1."""
    n, d, masks, times, d_fixed = mel_inputs(name)
    nchk = min(len(times), 3)
    cost, grad = onp.mellinger_cost_gradient(n, d, masks, times[:nchk], d_fixed[:nchk])
    ref_cost, ref_grad = MEL[f"{name}/cost_ref"][:nchk], MEL[f"{name}/grad_ref"][:nchk]
    tol = 1e-9 if n <= 10 else 5e-8
    assert np.abs(cost / ref_cost - 1).max() < tol
    # per trajectory: gradients of trajectories in one batch differ by orders of magnitude (a clamped 0.05 s segment)
    scale = np.maximum(np.abs(ref_grad).max(axis=1, keepdims=True), 1e-300)
    assert (np.abs(grad - ref_grad) / scale).max() < 10 * tol
    if times.shape[1] == 1:
        assert np.all(ref_grad == 0.0) and np.all(grad == 0.0)


@live_nl
@pytest.mark.parametrize("name", MEL_NAMES)
def test_committed_mellinger_fixture_is_what_the_library_produces(name):
    n, d, masks, times, d_fixed = mel_inputs(name)
    cost, grad = ref_linear.mellinger_cost_gradient(n, d, masks, times, d_fixed)
    assert np.array_equal(cost, MEL[f"{name}/cost_ref"]) and np.array_equal(grad, MEL[f"{name}/grad_ref"])
