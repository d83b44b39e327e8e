"""The reference's behaviour on RANK-DEFICIENT free systems behind the C ABI (MTG_FLAG_BASIC_SOLUTION, mtg_basic_solution_host).

LIN:365-378: SparseQR is rank-revealing, solveLinear() always returns a solution and `true`; under-constrained problems get a
BASIC solution.  The library's LDL^T sweep flags such trajectories, and with the flag solves them on the host
(csrc/mtg_basic.cpp).  The minimum cost is unique, the minimiser is not (the pivot order differs from Eigen's COLAMD order):
the tests compare what IS unique -- cost, constraints (checkPath), the stationarity of the quadratic -- with the reference
executed here (oracle/_ref), and the full solution where the system is regular.
CPU: the host routine through ctypes (the library loads without a GPU).  GPU: the flag through every pointer kind."""
import ctypes

import numpy as np
import pytest

import helpers
from oracle import ref_linear

needs_ref = pytest.mark.skipif(not ref_linear.available(), reason="compiled reference (oracle/_ref) not present")


def basic_one(n, k, dim, deriv, masks, times, d_fixed):
    """csrc/mtg_basic.cpp through ctypes: one trajectory (times [K], d_fixed [D][n_fixed]) -> (d_free [D][n_free], rank)."""
    from mav_trajectory_generation_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    fn = lib.mtg_basic_solution_one
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 4 + [ip, ip, ip, dp, dp, dp]
    h = n // 2
    m = np.array(masks, dtype=np.int32)
    off_f, off_p = np.zeros(k + 2, dtype=np.int32), np.zeros(k + 2, dtype=np.int32)
    for v in range(k + 1):
        nf = bin(masks[v]).count("1")
        off_f[v + 1], off_p[v + 1] = off_f[v] + nf, off_p[v] + h - nf
    t = np.ascontiguousarray(times, dtype=np.float64)
    f = np.ascontiguousarray(d_fixed, dtype=np.float64)
    out = np.full((dim, max(int(off_p[k + 1]), 1)), np.nan)
    rank = fn(h, k, dim, deriv, m.ctypes.data_as(ip), off_f.ctypes.data_as(ip), off_p.ctypes.data_as(ip), t.ctypes.data_as(dp),
              f.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out[:, :int(off_p[k + 1])], rank


@needs_ref
@pytest.mark.parametrize("n,k,dim,interior", [(10, 8, 3, 1), (8, 3, 2, 1), (12, 5, 3, 1), (10, 6, 4, 7), (6, 4, 1, 1)])
def test_regular_systems_full_rank_and_the_reference_solution(n, k, dim, interior):
    d = n // 2 - 1
    masks, times, d_fixed = helpers.reference_batch(4, k, n, dim, 77, masks=helpers.masks_ends_full(n, k, interior))
    _, fr_ref, _, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    for b in range(4):
        fr, rank = basic_one(n, k, dim, d, masks, times[b], d_fixed[b])
        assert rank == fr.shape[1]
        scale = np.abs(fr_ref[b]).max()
        assert np.abs(fr - fr_ref[b]).max() <= (1e-8 if n <= 10 else 1e-6) * scale


@needs_ref
@pytest.mark.parametrize("k,dim,masks", [(1, 3, [1, 1]), (1, 1, [1, 1]), (2, 3, [1, 1, 1]), (1, 2, [3, 1]), (2, 2, [1, 0, 1])])
def test_rank_deficient_systems_cost_and_stationarity(k, dim, masks):
    """Under-constrained problems (N = 10, snap): rank < n_free, the solution is basic (>= n_free - rank exact zeros), the
    quadratic is stationary in the free variables (R_PP d_P + R_PF d_F = 0: the system is consistent) and the cost
    0.5 d^T R d equals the reference's."""
    n, d = 10, 4
    rng = np.random.default_rng(5)
    nf = sum(bin(m).count("1") for m in masks)
    times = rng.uniform(0.8, 2.5, k)
    d_fixed = rng.uniform(-2.0, 2.0, (dim, nf))
    fr, rank = basic_one(n, k, dim, d, masks, times, d_fixed)
    npf = fr.shape[1]
    assert 0 < rank < npf
    assert int((fr[0] == 0.0).sum()) >= npf - rank
    _, r, nfix, nfree = ref_linear.m_and_r(n, d, masks, times, d_fixed)      # the reference's own R = M^T H M
    assert (nfix, nfree) == (nf, npf)
    _, _, cost_ref, _ = ref_linear.solve_batch(n, d, masks, times[None], d_fixed[None])
    cost = 0.0
    for dm in range(dim):
        dall = np.concatenate([d_fixed[dm], fr[dm]])
        grad = r[nf:, :] @ dall
        assert np.abs(grad).max() <= 1e-9 * max(np.abs(r).max() * np.abs(dall).max(), 1e-300)
        cost += 0.5 * dall @ r @ dall
    assert abs(cost - cost_ref[0]) <= 1e-9 * max(abs(cost_ref[0]), 1.0)


@pytest.fixture(scope="module")
def ctx():
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("k,dim,masks", [(1, 3, [1, 1]), (2, 3, [1, 1, 1]), (1, 2, [3, 1])])
@pytest.mark.parametrize("pointers", ["host", "host_backend", "device_aos", "device_soa"])
def test_basic_solution_flag_through_the_c_abi(ctx, k, dim, masks, pointers):
    """mtg_solve_linear[_status] with MTG_FLAG_BASIC_SOLUTION returns MTG_OK on an under-constrained batch like the reference's
    solveLinear() returns true; without the flag the same call reports MTG_ERR_SINGULAR.  Constraints and cost as the
    reference's (checkPath 1e-6, TOPT:116; cost to 1e-9)."""
    import torch
    import mav_trajectory_generation_amd as m
    n, d, bsz = 10, 4, 37
    rng = np.random.default_rng(11 + k)
    nf = sum(bin(x).count("1") for x in masks)
    times = rng.uniform(0.8, 2.5, (bsz, k))
    d_fixed = rng.uniform(-2.0, 2.0, (bsz, dim, nf))
    _, _, cost_ref, _ = ref_linear.solve_batch(n, d, masks, times, d_fixed)
    plan = m.Plan(ctx, n, dim, k, d, masks)
    if pointers.startswith("host"):
        hb = pointers == "host_backend"
        with pytest.raises(m.MtgError) as e:
            plan.solve_host(times[:20], d_fixed[:20], host_backend=hb)
        assert e.value.code == -3
        co, fr, cost = plan.solve_host(times, d_fixed, host_backend=hb and bsz <= 64, basic_solution=True)
    else:
        layout = pointers.split("_")[1]
        t = torch.from_numpy(times).cuda()
        f = torch.from_numpy(d_fixed).cuda()
        if layout == "soa":
            t, f = t.t().contiguous(), f.permute(1, 2, 0).contiguous()
        plan.solve(t, f, layout=layout, want_free=True, want_cost=True)
        with pytest.raises(m.MtgError) as e:
            ctx.sync()
        assert e.value.code == -3
        st = torch.zeros(bsz, dtype=torch.int32, device="cuda")
        co, fr, cost = plan.solve(t, f, layout=layout, want_free=True, want_cost=True, traj_status=st, basic_solution=True)
        ctx.sync()                                           # the flag consumed the status word: nothing left to report
        assert int((st & 2).ne(0).sum()) > 0                 # WHICH trajectories were basic stays visible
        co, cost = co.cpu().numpy(), cost.cpu().numpy()
        fr = fr.cpu().numpy() if layout == "aos" else fr.permute(2, 0, 1).cpu().numpy()
    assert np.isfinite(co).all() and np.isfinite(fr).all()
    assert helpers.check_path(masks, times, d_fixed, co) < 1e-6
    assert np.abs(cost - cost_ref).max() <= 1e-9 * max(np.abs(cost_ref).max(), 1.0)
    plan.close()


@pytest.mark.gpu
def test_basic_solution_flag_leaves_regular_batches_alone(ctx):
    """A regular batch: the flag changes nothing (bit-identical outputs), bad segment times are still reported."""
    import torch
    import mav_trajectory_generation_amd as m
    masks = m.ends_full_masks(10, 8)
    plan = m.Plan(ctx, 10, 3, 8, 4, masks)
    t, f = m.random_waypoint_batch(500, 8, 3, 10, masks, seed=3, device="cuda", layout="soa")
    a, _, ca = plan.solve(t, f, layout="soa", want_cost=True)
    b, _, cb = plan.solve(t, f, layout="soa", want_cost=True, basic_solution=True)
    ctx.sync()
    assert torch.equal(a, b) and torch.equal(ca, cb)
    t[3, 17] = -1.0
    with pytest.raises(m.MtgError) as e:
        plan.solve(t, f, layout="soa", basic_solution=True)
    assert e.value.code == -2
    ctx.sync()
    plan.close()


@pytest.mark.gpu
@needs_ref
def test_basic_solution_host_entry(ctx):
    import mav_trajectory_generation_amd as m
    from mav_trajectory_generation_amd import _lib
    plan = m.Plan(ctx, 10, 2, 1, 4, [1, 1])
    t = np.array([1.7])
    f = np.array([[0.5, -1.0], [2.0, 0.25]])
    fr = np.empty((2, plan.n_free))
    rank = ctypes.c_int32(-1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert _lib.load().mtg_basic_solution_host(plan.handle, p(t), p(f), p(fr), ctypes.byref(rank)) == 0
    want, r = basic_one(10, 1, 2, 4, [1, 1], t, f)
    assert rank.value == r < plan.n_free and np.array_equal(fr, want)
    assert _lib.load().mtg_basic_solution_host(plan.handle, p(np.array([0.0])), p(f), p(fr), None) == -2
    plan.close()
