"""CPU tests of the DEVICE lane code (mtg_lane.h) compiled for the host by tests/host_emu.cpp: same
arithmetic as the HIP kernels (twisted block-LDL^T, unit-time tables), checked against the oracle and the
mpmath truth.  The emulation is test infrastructure; the product has no CPU path."""
import os

import numpy as np
import pytest

import helpers
from oracle import oracle_np as onp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_linear_golden.npz"))


def case(name):
    pre = name + "/"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def tol_for(n, d):
    # float64 evaluation of the reference's formulas is itself only this accurate (see test_oracle.py)
    if n == 12 and d < n // 2 - 1:
        return 1e-5   # e.g. the yaw instantiation of test_feasibility.cpp:109-113: cond(R_PP) ~ 1e8
    if n == 12 or d < n // 2 - 1:
        return 5e-7
    return 1e-9


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_emulated_kernel_matches_oracle_on_golden(host_emu, name, mode):
    c = case(name)
    n, d = int(c["n"]), int(c["d"])
    masks = [int(m) for m in c["masks"]]
    k = c["times"].shape[1]
    dim = c["d_fixed"].shape[1]
    rc, co, fr, cost, st = helpers.emu_run(host_emu, n, dim, k, d, masks, c["times"], c["d_fixed"], mode)
    if mode in (1, 3) and rc == -2:
        pytest.skip("no specialised variant for this shape")
    assert rc == 0 and st == 0
    assert helpers.poly_relerr(co, c["coeffs_lit"]) < tol_for(n, d)
    if "coeffs_mp" in c:
        # the kernel algorithm is closer to the truth than the literal evaluation
        assert helpers.poly_relerr(co, c["coeffs_mp"]) < (1e-7 if n == 12 else 1e-11)
        if c["d_free_mp"].size:
            assert np.abs(fr - c["d_free_mp"]).max() <= 1e-7 * max(1.0, np.abs(c["d_free_mp"]).max())
        assert np.allclose(cost, c["cost_mp"], rtol=1e-7)
    assert helpers.check_path(masks, c["times"], c["d_fixed"], co) < 1e-6


def test_update_from_free_matches_solve(host_emu):
    c = case("config2")
    masks = [int(m) for m in c["masks"]]
    rc, co, fr, cost, st = helpers.emu_run(host_emu, 10, 3, 8, 4, masks, c["times"], c["d_fixed"], 0)
    rc2, co2, _, cost2, st2 = helpers.emu_run(host_emu, 10, 3, 8, 4, masks, c["times"], c["d_fixed"], 2, d_free_in=fr)
    assert rc2 == 0 and st2 == 0
    assert helpers.poly_relerr(co2, co) < 1e-13
    assert np.allclose(cost, cost2, rtol=1e-12)


def test_bad_time_and_singular_flags(host_emu):
    c = case("config2")
    masks = [int(m) for m in c["masks"]]
    t = c["times"].copy()
    t[3, 2] = 0.0
    rc, _, _, _, st = helpers.emu_run(host_emu, 10, 3, 8, 4, masks, t, c["d_fixed"], 0)
    assert st & 1  # MTG_FLAG_BAD_TIME (LIN:297 CHECK_GT)


@pytest.mark.parametrize("n,d,k,dim,masks", [
    (10, 4, 6, 3, [31, 1, 3, 1, 5, 9, 31]),      # ragged interior masks
    (10, 4, 5, 5, None),                          # D > 4 -> dimension chunks
    (10, 4, 3, 2, [3, 1, 1, 7]),                  # free end-vertex derivatives
    (2, 0, 3, 2, None), (4, 1, 3, 2, None), (6, 2, 4, 3, None),
    (12, 5, 7, 4, None),
])
def test_emulated_kernel_edge_shapes(host_emu, n, d, k, dim, masks):
    masks, times, d_fixed = helpers.reference_batch(6, k, n, dim, 4242, masks)
    c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, times, d_fixed)
    rc, co, fr, cost, st = helpers.emu_run(host_emu, n, dim, k, d, masks, times, d_fixed, 0)
    assert rc == 0 and st == 0
    assert helpers.poly_relerr(co, c_lit) < tol_for(n, d)
    assert helpers.check_path(masks, times, d_fixed, co) < 1e-6


def test_randomised_shapes_and_masks(host_emu):
    """Seeded sweep over ragged per-vertex masks (position always fixed), K, D, N: the generic code path of the
    kernel (run-time masks, dimension chunks, workspace) against the literal oracle + checkPath."""
    rng = np.random.default_rng(20240924)
    for trial in range(60):
        n = int(rng.choice([4, 6, 8, 10, 10, 10, 12]))
        h = n // 2
        d = h - 1 if rng.random() < 0.8 else int(rng.integers(max(1, h - 2), h))
        k = int(rng.integers(1, 10))
        dim = int(rng.integers(1, 6))
        masks = [int(1 | (rng.integers(0, 1 << h) if rng.random() < 0.6 else 0)) for _ in range(k + 1)]
        if rng.random() < 0.5:
            masks[0] = masks[-1] = (1 << h) - 1
        m2, times, d_fixed = helpers.reference_batch(3, k, n, dim, 9000 + trial, masks)
        c_lit, f_lit, j_lit = onp.solve_batch(n, d, masks, times, d_fixed)
        rc, co, fr, cost, st = helpers.emu_run(host_emu, n, dim, k, d, masks, times, d_fixed, 0)
        # rank-deficient free systems (too few constraints for the null space of the cost, e.g. one segment with
        # only the two end positions fixed): the reference's rank-revealing QR returns *a* minimiser, the kernel
        # reports MTG_FLAG_SINGULAR or solves a numerically singular system -- only well-posed cases are compared
        opt = onp.PolynomialOptimization(n, dim)
        verts = [onp.Vertex(dim) for _ in range(k + 1)]
        col = 0
        for v in range(k + 1):
            for p in range(h):
                if (masks[v] >> p) & 1:
                    verts[v].add_constraint(p, d_fixed[0, :, col])
                    col += 1
        opt.setup_from_vertices(verts, times[0], d)
        if opt.n_free:
            nf = opt.n_fixed
            ev = np.linalg.eigvalsh(opt.construct_r()[nf:, nf:])
            if ev.min() < 1e-11 * ev.max():
                assert rc == 0
                continue
        assert rc == 0 and st == 0, (trial, n, d, k, dim, masks)
        tol = 1e-8 if (n <= 10 and d == h - 1) else 1e-4
        assert helpers.poly_relerr(co, c_lit) < tol, (trial, n, d, k, dim, masks)
        assert helpers.check_path(masks, times, d_fixed, co) < 1e-6, (trial, n, d, k, dim, masks)
        if j_lit.min() > 1e-9:
            assert np.allclose(cost, j_lit, rtol=1e-5), (trial, n, d, k, dim, masks)


def test_extreme_time_ratios_stay_closer_to_truth_than_the_literal_route(host_emu):
    """Segment times log-uniform over three decades (ratios up to 1e3 between neighbouring segments): the problem
    itself becomes ill-conditioned; the kernel's formulation (no inversion of A(T), scaled constants) must stay
    orders of magnitude closer to the 50-digit solution than the literal float64 evaluation of the reference."""
    from oracle import oracle_mp as omp
    rng = np.random.default_rng(5)
    masks, times, d_fixed = helpers.reference_batch(6, 8, 10, 3, 77)
    times = np.exp(rng.uniform(np.log(0.05), np.log(50.0), times.shape))
    c_mp, _, _ = omp.solve_batch(10, 4, masks, times, d_fixed)
    c_lit, _, _ = onp.solve_batch(10, 4, masks, times, d_fixed)
    rc, co, _, _, st = helpers.emu_run(host_emu, 10, 3, 8, 4, masks, times, d_fixed, 1)
    assert rc == 0 and st == 0
    e_kernel, e_lit = helpers.poly_relerr(co, c_mp), helpers.poly_relerr(c_lit, c_mp)
    assert e_kernel < 1e-5 and e_kernel < 1e-2 * e_lit


@pytest.mark.parametrize("n,k,mi", [(8, 4, 1), (8, 8, 1), (10, 4, 1), (10, 8, 1), (12, 4, 1), (12, 8, 1),
                                    (10, 8, 3), (10, 8, 7), (10, 5, 3), (12, 5, 1)])     # + other interior masks, odd K
def test_shared_workspace_factor_store_emulation(host_emu, n, k, mi):
    """The dimension-in-lane form with shared step storage, every step through the lane-coalesced workspace: three dimension
    lanes of a trajectory keep ONE copy of a step's matrix between them -- since round 4 the LDL^T factor of the pivot block
    (MtgCfg::kFS), with U rebuilt from the table at back-substitution time -- and must reproduce the generic lane code
    (which keeps G = Dtilde^-1 U per lane) to round-off, and the C++ port / the 50-digit oracle to the usual tolerance."""
    import ctypes
    from oracle import cpu_ref
    dp = ctypes.POINTER(ctypes.c_double)
    host_emu.mtg_emu_run_shared.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, dp, dp, dp,
                                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    host_emu.mtg_emu_run_shared.restype = ctypes.c_int
    d, dim, bsz = n // 2 - 1, 3, 24
    masks = helpers.masks_ends_full(n, k, mi)
    _, times, fixed = helpers.reference_batch(bsz, k, n, dim, 900 + n + k, masks)
    times, fixed = np.ascontiguousarray(times), np.ascontiguousarray(fixed)
    co = np.zeros((bsz, k, dim, n))
    st, fs = ctypes.c_int(0), ctypes.c_int(-1)
    rc = host_emu.mtg_emu_run_shared(n, k, mi, bsz, times.ctypes.data_as(dp), fixed.ctypes.data_as(dp), co.ctypes.data_as(dp),
                                     ctypes.byref(st), ctypes.byref(fs))
    assert rc == 0 and st.value == 0
    assert fs.value == 1        # this build keeps factors
    rc0, co0, _, _, st0 = helpers.emu_run(host_emu, n, dim, k, d, masks, times, fixed, 0, want_cost=False)
    assert rc0 == 0 and st0 == 0
    # same algorithm up to the back-substitution's association: g - G x  vs  g - (L D L^T)^-1 (U x)
    assert helpers.poly_relerr(co, co0) < (2e-9 if n == 12 else 2e-11)
    port = cpu_ref.solve_batch(n, d, masks, times, fixed)[0]
    assert helpers.poly_relerr(co, port) < (5e-7 if n == 12 else 1e-9)
    from oracle import oracle_mp
    truth = np.asarray(oracle_mp.solve_batch(n, d, masks, times[:3], fixed[:3])[0], dtype=np.float64)
    e_fs, e_g = helpers.poly_relerr(co[:3], truth), helpers.poly_relerr(co0[:3], truth)
    assert e_fs < (1e-7 if n == 12 else 1e-11)
    assert e_fs < 20 * e_g + 1e-13      # not materially further from the truth than the G form
    assert helpers.check_path(masks, times, fixed, co) < 1e-6


@pytest.mark.parametrize("n,k", [(8, 8), (10, 8), (10, 16), (12, 8), (12, 16)])
def test_scaled_variable_chain_generic_and_specialised_kernels_agree(host_emu, n, k):
    """Round 6, the scaled-variable chain (mtg_lane.h): the sweep carries the Schur complement in the current segment's scaling and
    converts it with the factors rho^(2d - 1 - p - q), rho = T / T'.  The generic kernels (run-time derivative) and the specialised ones
    (static / rolled) must form those factors by the SAME power table -- a one-ulp difference in them is round-off x cond(R_PP) in the
    solution (2e-10 on the worst of 125k config-2 trajectories when the generic kernel multiplied kappa sigma^m instead) -- and both
    stay at round-off x conditioning of the 50-digit solution on uneven segment times."""
    from oracle import oracle_mp as omp
    d = n // 2 - 1
    masks, times, d_fixed = helpers.reference_batch(16, k, n, 3, 4242 + n + k)
    rng = np.random.default_rng(n * 100 + k)
    times = times * np.exp(rng.uniform(np.log(0.3), np.log(3.0), times.shape))      # neighbouring segments up to 10x apart
    rc0, c_gen, fr0, j0, st0 = helpers.emu_run(host_emu, n, 3, k, d, masks, times, d_fixed, 0)
    assert rc0 == 0 and st0 == 0
    for mode in (1, 3):
        rc, c_sp, fr, j, st = helpers.emu_run(host_emu, n, 3, k, d, masks, times, d_fixed, mode)
        if rc == -2:
            continue
        assert rc == 0 and st == 0
        assert helpers.poly_relerr(c_sp, c_gen) < (1e-12 if n <= 10 else 1e-10), (mode, helpers.poly_relerr(c_sp, c_gen))
        assert np.allclose(j, j0, rtol=1e-10)
    c_mp = np.asarray(omp.solve_batch(n, d, masks, times, d_fixed)[0], dtype=np.float64)
    assert helpers.poly_relerr(c_gen, c_mp) < {8: 1e-12, 10: 1e-10, 12: 1e-6}[n]
