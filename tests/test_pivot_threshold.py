"""The rank decision of the solve (LIN:365-378: the reference's SparseQR is rank-revealing, solveLinear() returns a basic solution).

Two layers:
  * STRUCTURE (csrc/mtg_abi.hip, structural_null_dim / mtg_structural_rank_deficiency): R_PP's null space is the polynomials of
    degree < d over the whole trajectory on which every fixed slot vanishes -- a property of the constraint pattern.  Computed
    once per plan; every trajectory of a deficient plan is flagged by every launch route.  Checked here against the rank the
    library's own pivoted QR (csrc/mtg_basic.cpp: Eigen's threshold) finds on random values, over hundreds of random patterns,
    Birkhoff-type ones (a derivative fixed without the lower ones) included.
  * PIVOTS (csrc/mtg_lane.h, mtg_pivot_tau / mtg_ldl): on regular plans a pivot counts as zero when
    d_j <= 20 (n_free + n_free) eps x R_PP[j][j] -- the form of SparseQR's default threshold relative to the variable's own
    diagonal (rounds 1-4: the sign of d_j).  This is a guard (lost digits, NaN, T beyond float64), NOT a rank test: on chains
    of free vertices a structurally zero pivot comes out anywhere between 1e-12 and 1e0 of the diagonal, of either sign
    (profiles/r05_pivot_ratio_study.txt), overlapping the legitimate pivots of regular ill-conditioned problems -- which is why
    the decision moved to the structure.  Checked here: no false positives on regular ill-conditioned problems under BOTH
    associations of the Schur update (MTG_PARTIAL_ALL = 0 / 1).
GPU: deficient structures inside chains of 8 ... 50 segments through every launch route, with and without
MTG_FLAG_BASIC_SOLUTION, against the reference's cost.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers
from oracle import ref_linear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc")
needs_ref = pytest.mark.skipif(not ref_linear.available(), reason="compiled reference (oracle/_ref) not present")


def build_emu(partial_all):
    name = "libmtg_host_emu_partial.so" if partial_all else "libmtg_host_emu.so"
    so, src = os.path.join(ROOT, "tests", name), os.path.join(ROOT, "tests", "host_emu.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("mtg_lane.h", "mtg_tables.inc", "mtg_variants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC"] + (["-DMTG_PARTIAL_ALL=1"] if partial_all else []) + ["-o", so, src])
    lib = ctypes.CDLL(so)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib.mtg_emu_run.argtypes = [ctypes.c_int] * 4 + [ip, ctypes.c_longlong, dp, dp, dp, dp, dp, ctypes.c_int, ip]
    lib.mtg_emu_run.restype = ctypes.c_int
    return lib


@pytest.fixture(scope="module")
def emus():
    return {0: build_emu(False), 1: build_emu(True)}


def random_problem(rng, n, k, dim, masks, tlo=0.5, thi=3.0):
    nf = sum(bin(m).count("1") for m in masks)
    return rng.uniform(tlo, thi, (1, k)), rng.uniform(-2.0, 2.0, (1, dim, nf))


def structures(n, k):
    """(name, masks, constraints that act on the cost's null space) -- positions / velocities at distinct instants are
    independent conditions on a polynomial of degree < d."""
    out = [("ends_position_only", [1] + [0] * (k - 1) + [1], 2)]
    if k >= 2:
        mid = [0] * (k - 1)
        mid[(k - 1) // 2] = 1
        out.append(("ends_and_one_interior_position", [1] + mid + [1], 3))
        out.append(("start_position_velocity_only", [3] + [0] * k, 2))
    return out


def structural(n, k, d, masks):
    from mav_trajectory_generation_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    arr = (ctypes.c_uint32 * (k + 1))(*masks)
    return lib.mtg_structural_rank_deficiency(n, k, d, arr)


@pytest.mark.parametrize("n", [8, 10, 12])
@pytest.mark.parametrize("k", [1, 2, 3, 8, 16, 32, 50])
def test_structural_rank_deficiency_of_under_constrained_chains(n, k):
    d = n // 2 - 1
    for name, masks, n_constraints in structures(n, k):
        assert structural(n, k, d, masks) == max(0, d - n_constraints), (name, n, k)
    assert structural(n, k, d, helpers.masks_ends_full(n, k, 1)) == 0
    assert structural(n, k, 0, [0] * (k + 1)) == 0           # d = 0: the cost is the integral of p^2, positive definite


def test_structural_rank_agrees_with_the_pivoted_qr_on_random_patterns():
    """Random constraint patterns (any subset of slots per vertex: Hermite- and Birkhoff-type), random values: n_free - rank of
    the library's rank-revealing QR (csrc/mtg_basic.cpp, Eigen's threshold, LIN:365-378) == the structural number."""
    from test_basic_solution import basic_one
    rng = np.random.default_rng(2024)
    seen_deficient = 0
    for trial in range(300):
        n = int(rng.choice([4, 6, 8, 10]))
        h = n // 2
        d = int(rng.integers(1, h))
        k = int(rng.integers(1, 6))
        sparse = rng.random() < 0.6          # mostly few constraints: that is where the deficient patterns are
        masks = [int(rng.integers(0, 1 << h)) for _ in range(k + 1)]
        if sparse:
            masks = [m & int(rng.integers(0, 1 << h)) & int(rng.integers(0, 1 << h)) for m in masks]
        nf = sum(bin(m).count("1") for m in masks)
        if nf == (k + 1) * h:
            continue
        times = rng.uniform(0.7, 2.0, k)
        d_fixed = rng.uniform(-1.0, 1.0, (1, nf))
        fr, rank = basic_one(n, k, 1, d, masks, times, d_fixed)
        want = structural(n, k, d, masks)
        assert fr.shape[1] - rank == want, (trial, n, d, k, masks, rank, fr.shape[1], want)
        seen_deficient += want > 0
    assert seen_deficient >= 40


@pytest.mark.parametrize("n,d,k,masks,tlo,thi", [
    (12, 2, 5, [3, 1, 1, 1, 1, 3], 0.5, 3.0),                 # the yaw pattern of R/test/test_feasibility.cpp:109-113: cond 1.5e-8
    (12, 5, 32, None, 0.2, 5.0), (12, 5, 50, None, 0.5, 3.0),      # (N = 12 with ratios of 400 is beyond float64: negative pivots)
    (10, 2, 5, None, 0.5, 3.0), (10, 3, 9, None, 0.05, 20.0), (10, 4, 100, None, 0.5, 3.0), (10, 4, 16, None, 0.05, 20.0),
    (8, 3, 100, None, 0.05, 20.0), (8, 1, 6, None, 0.5, 3.0), (6, 2, 7, [7, 1, 0, 1, 2, 0, 1, 7], 0.5, 3.0),
])
def test_regular_ill_conditioned_problems_are_not_flagged(emus, n, d, k, masks, tlo, thi):
    """No false positives: d_j / R_PP[j][j] >= 1 / cond(R_PP) stays orders of magnitude above tau ~ 1e-13 on every well-posed
    problem the reference's tests and BASELINE.json's configs contain, segment-time ratios of 400 included."""
    if masks is None:
        masks = helpers.masks_ends_full(n, k, 1)
    rng = np.random.default_rng(7 * n + k)
    for trial in range(6):
        times, d_fixed = random_problem(rng, n, k, 1, masks, tlo, thi)
        if thi / tlo > 20:                       # short and long segments next to each other
            times[0, ::2] = tlo * (1.0 + 0.1 * trial)
        for assoc, lib in emus.items():
            rc, co, _, _, st = helpers.emu_run(lib, n, 1, k, d, masks, times, d_fixed)
            assert rc == 0 and st == 0 and np.isfinite(co).all(), (assoc, trial, st)


# ---- GPU: the same structures through the library --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    c.set_option("coop", -1)
    yield c
    c.close()


def costly_structures(n, k):
    """Under-constrained patterns whose minimum cost is NOT zero: the trajectory ends fix the position and the highest slot
    (derivative h - 1 = d: no condition on the cost's null space, but a non-zero value forces a non-zero cost); every other slot
    free, optionally one interior position.  (With positions only, a polynomial of degree < d interpolates them: cost 0, and a
    relative comparison of two round-off-level numbers says nothing.)"""
    h = n // 2
    top = 1 << (h - 1)
    out = [("ends_position_and_top", [1 | top] + [0] * (k - 1) + [1 | top], 2)]
    if k >= 2:
        mid = [0] * (k - 1)
        mid[(k - 1) // 2] = 1
        out.append(("ends_position_and_top_one_interior_position", [1 | top] + mid + [1 | top], 3))
    return out


def pinned_masks(n, k, d, masks):
    """Test-side restatement of the library's choice of pinned slots (csrc/mtg_abi.hip, structural_null_dim): the lowest free
    slots (vertex, derivative < d) whose functionals p -> p^(q)(t_v) complete the fixed ones to a basis on P_(d-1)."""
    h = n // 2
    tv = np.cumsum([0.0] + [0.35 + ((v + 1) * 0.6180339887498949) % 1 for v in range(1, k + 1)]) / k

    def row(v, q):
        r = np.zeros(d)
        for m in range(q, d):
            c = 1.0
            for i in range(q):
                c *= m - i
            r[m] = c * tv[v] ** (m - q)
        return r
    rows = [row(v, q) for v in range(k + 1) for q in range(min(h, d)) if (masks[v] >> q) & 1]
    a = np.array(rows) if rows else np.zeros((0, d))
    rank = np.linalg.matrix_rank(a, tol=1e-9) if len(rows) else 0
    out = list(masks)
    for v in range(k + 1):
        for q in range(min(h, d)):
            if rank >= d or (masks[v] >> q) & 1:
                continue
            a2 = np.vstack([a, row(v, q)])
            if np.linalg.matrix_rank(a2, tol=1e-9) > rank:
                a, rank = a2, rank + 1
                out[v] |= 1 << q
    return out


def assert_cost_matches(n, d, masks, times, d_fixed, cost, cost_ref, tol):
    """Cost against the reference's; a trajectory beyond the tolerance is arbitrated by the 50-digit solve of the PINNED (regular)
    system, whose minimum equals the deficient system's: accepted only when the HIP cost is within `tol` of it (seen on
    50-segment chains of free vertices: the reference's QR 3.5e-8 off on one of 21 trajectories, the HIP path 1e-11)."""
    rel = np.abs(cost - cost_ref) / np.abs(cost_ref)
    over = np.nonzero(~(rel <= tol))[0]
    assert len(over) <= max(2, len(rel) // 10), rel
    if len(over):
        from oracle import oracle_mp
        k = times.shape[1]
        pm = pinned_masks(n, k, d, masks)
        h = n // 2
        cols, src = [], 0
        for v in range(k + 1):
            for q in range(h):
                if (pm[v] >> q) & 1:
                    cols.append(src if (masks[v] >> q) & 1 else -1)
                    src += (masks[v] >> q) & 1
        for b in over:
            sd = np.zeros((d_fixed.shape[1], len(cols)))
            for j, c in enumerate(cols):
                if c >= 0:
                    sd[:, j] = d_fixed[b][:, c]
            truth = oracle_mp.solve(n, d, pm, times[b], sd)[2]
            assert abs(cost[b] - truth) <= tol * abs(truth), (int(b), cost[b], cost_ref[b], truth)


def assert_same_minimiser_up_to_the_null_space(n, d, masks, times, co, fr, co_ref, fr_ref, null_dim):
    """The relation between the library's basic solution and the compiled reference's (oracle/_ref) on a rank-deficient problem,
    coefficient level.  Both are minimisers (same cost, same constraints: asserted by the caller), both have null_dim exactly-zero
    free variables -- at DIFFERENT slots: the library pins the LOWEST free slots (vertex, derivative < d) that complete the fixed
    ones; oracle/_ref's QR drops the columns it meets last in ITS column order (the stand-in's minimum-degree
    order; real Eigen: COLAMD's -- not reproducible in this image), here the top derivatives < d of the last interior vertex.
    Hence the coefficients differ, by an element of the cost's null space: ONE polynomial of degree < d over the whole
    trajectory.  Asserted: the d-th derivative of the difference vanishes on every segment (relative to the solution's own d-th
    derivative); the difference is not small (the two solutions ARE different minimisers)."""
    k, h = times.shape[1], n // 2
    slots = [(v, q) for v in range(k + 1) for q in range(h) if not (masks[v] >> q) & 1]
    zeros_lib = [slots[i] for i in np.nonzero(fr[0, 0] == 0.0)[0]]
    zeros_ref = [slots[i] for i in np.nonzero(fr_ref[0, 0] == 0.0)[0]]
    assert len(zeros_lib) == null_dim and len(zeros_ref) == null_dim, (zeros_lib, zeros_ref)
    assert all(q < d for (_, q) in zeros_lib + zeros_ref)                    # only slots the cost's null space can move
    assert zeros_lib == sorted(zeros_lib) and zeros_lib[0][0] <= 1 and zeros_ref[-1][0] >= k - 1, (zeros_lib, zeros_ref)
    diff = co - co_ref
    worst = scale = 0.0
    for s in range(k):
        for tt in np.linspace(0.0, 1.0, 9):
            t = tt * times[:, s, None]
            worst = max(worst, np.abs(helpers.evaluate(diff[:, s], t, d)).max())
            scale = max(scale, np.abs(helpers.evaluate(co_ref[:, s], t, d)).max())
    # a zero-cost direction: the difference's own cost is (worst / scale)^2 of the solution's, i.e. <= 1e-9 (N <= 10) / 1e-6 (N = 12)
    # -- the caller's cost tolerance; measured 1e-9 ... 8e-7 (N <= 10, K = 8 ... 50), 4e-6 (N = 12) for worst / scale
    assert worst <= (3e-5 if n <= 10 else 1e-3) * scale, (worst, scale)
    assert np.abs(diff).max() > 1e-3 * np.abs(co_ref).max()                    # ... and not the same point of the solution set


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("n,k", [(10, 8), (10, 16), (10, 32), (10, 50), (8, 16), (8, 50), (12, 8)])
@pytest.mark.parametrize("which", ["ends_position_and_top", "ends_position_and_top_one_interior_position"])
def test_rank_deficient_long_chains_through_every_launch_route(ctx, n, k, which):
    """Free interior vertices, ends fixed to order 0 (+ the highest slot): single launch (device pointers, both layouts), host
    pointers, the host backend, a queue (mtg_solve_linear_sequence) and a merged request (mtg_multi_*): EVERY trajectory of a
    structurally deficient plan is flagged by every route; with MTG_FLAG_BASIC_SOLUTION the call returns MTG_OK, the solution
    comes from the plan's shadow (the pinned, regular system) with the reference's cost -- 1e-9 for N <= 10 on chains of 50
    free vertices (the dense pivoted QR of rounds 3-4: 1e-5 there) -- and constraints (checkPath 1e-6, TOPT:116)."""
    import torch
    import mav_trajectory_generation_amd as m
    d, dim, bsz = n // 2 - 1, 3, 21
    name, masks, n_constraints = [s for s in costly_structures(n, k) if s[0] == which][0]
    deficient = n_constraints < d
    rng = np.random.default_rng(k + n)
    nf = sum(bin(x).count("1") for x in masks)
    times, d_fixed = rng.uniform(0.8, 2.5, (bsz, k)), rng.uniform(-2.0, 2.0, (bsz, dim, nf))
    co_ref, fr_ref, cost_ref, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed, nthreads=ref_linear.hardware_threads())
    cost_tol = 1e-9 if n <= 10 else 1e-6
    plan = m.Plan(ctx, n, dim, k, d, masks)
    assert plan.rank_deficiency == max(0, d - n_constraints)
    for layout in ("aos", "soa"):
        t, f = torch.from_numpy(times).cuda(), torch.from_numpy(d_fixed).cuda()
        if layout == "soa":
            t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
        st = torch.full((bsz,), 9, dtype=torch.int32, device="cuda")
        plan.solve(t, f, layout=layout, traj_status=st)
        if deficient:
            with pytest.raises(m.MtgError) as e:
                ctx.sync()
            assert e.value.code == -3
        else:
            ctx.sync()
        assert (((st.cpu().numpy() & 2) != 0) == deficient).all()           # the whole batch shares the structure
        # a queue of three batches of the same plan: the context's word reports it
        sets = [(t, f, torch.zeros((bsz, k, dim, n), dtype=torch.float64, device="cuda")) for _ in range(3)]
        plan.solve_sequence(sets, layout=layout)
        if deficient:
            with pytest.raises(m.MtgError) as e:
                ctx.sync()
            assert e.value.code == -3
        else:
            ctx.sync()
        # ... and a merged request
        req = m.MultiSolve(ctx, [dict(plan=plan, times=t, d_fixed=f, layout=layout) for _ in range(2)])
        req.solve()
        if deficient:
            with pytest.raises(m.MtgError) as e:
                ctx.sync()
            assert e.value.code == -3
        else:
            ctx.sync()
        req.close()
        # the reference's behaviour: MTG_OK, a basic solution
        st2 = torch.zeros(bsz, dtype=torch.int32, device="cuda")
        co, fr, cost = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, traj_status=st2, basic_solution=True)
        ctx.sync()
        assert (((st2.cpu().numpy() & 2) != 0) == deficient).all()
        co, cost = co.cpu().numpy(), cost.cpu().numpy()
        fr = fr.cpu().numpy() if layout == "aos" else fr.permute(2, 0, 1).cpu().numpy()
        assert np.isfinite(co).all() and helpers.check_path(masks, times, d_fixed, co) < 1e-6
        assert_cost_matches(n, d, masks, times, d_fixed, cost, cost_ref, cost_tol)
        if deficient:       # basic: (at least) as many free variables as the rank is short are exactly zero
            assert int((fr[0, 0] == 0.0).sum()) >= plan.rank_deficiency
            assert_same_minimiser_up_to_the_null_space(n, d, masks, times, co, fr, co_ref, fr_ref, plan.rank_deficiency)
        # round 6: the flag through the ASYNCHRONOUS batched entries -- a queue and a merged request run the plan's shadow
        sets = [(t, f, torch.full((bsz, k, dim, n), float("nan"), dtype=torch.float64, device="cuda")) for _ in range(3)]
        plan.solve_sequence(sets, layout=layout, basic_solution=True)
        ctx.sync()                                                          # MTG_OK: nothing flagged
        for (_, _, c3) in sets:
            assert np.abs(c3.cpu().numpy() - co).max() <= 1e-9 * np.abs(co).max()
        req = m.MultiSolve(ctx, [dict(plan=plan, times=t, d_fixed=f, layout=layout) for _ in range(2)], want_free=True, want_cost=True,
                           basic_solution=True)
        outs = req.solve()
        ctx.sync()
        for (c4, f4, j4) in outs:
            f4 = f4.cpu().numpy() if layout == "aos" else f4.permute(2, 0, 1).cpu().numpy()
            assert np.abs(c4.cpu().numpy() - co).max() <= 1e-9 * np.abs(co).max()
            assert np.abs(j4.cpu().numpy() / cost - 1).max() <= 1e-9
            assert np.array_equal(f4 == 0.0, fr == 0.0) and np.abs(f4 - fr).max() <= 1e-9 * np.abs(fr).max()
        req.close()
    for hb in (False, True):
        if deficient:
            with pytest.raises(m.MtgError) as e:
                plan.solve_host(times, d_fixed, host_backend=hb)
            assert e.value.code == -3
        co, fr, cost = plan.solve_host(times, d_fixed, host_backend=hb, basic_solution=True)
        assert helpers.check_path(masks, times, d_fixed, co) < 1e-6
        assert_cost_matches(n, d, masks, times, d_fixed, cost, cost_ref, cost_tol)
        if deficient:
            assert int((fr[0, 0] == 0.0).sum()) >= plan.rank_deficiency
    plan.close()


@pytest.mark.gpu
def test_basic_solution_call_keeps_the_contexts_status_word(ctx):
    """a device-pointer MTG_FLAG_BASIC_SOLUTION call has its own status word -- the BAD_TIME flag an EARLIER
    asynchronous launch left in the context is still reported by the next mtg_context_sync."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(64, 8, 3, 10, masks, seed=2, device="cuda", layout="aos")
    tb = t.clone()
    tb[5, 3] = -1.0
    plan.solve(tb, f)                                            # asynchronous: leaves BAD_TIME in the context's word
    co, _, _ = plan.solve(t, f, basic_solution=True)             # synchronous, regular batch: MTG_OK, own status word
    assert torch.isfinite(co).all()
    with pytest.raises(m.MtgError) as e:
        ctx.sync()
    assert e.value.code == -2
    ctx.sync()
    # a bad time in the flagged call itself is the call's own result, and nothing of it stays behind in the context
    with pytest.raises(m.MtgError) as e:
        plan.solve(tb, f, basic_solution=True)
    assert e.value.code == -2
    ctx.sync()
    plan.close()
