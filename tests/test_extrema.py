"""Row N4: magnitude extrema + feasibility time scaling.

CPU part: the oracle restatement (oracle/oracle_extrema.py) is pinned against the reference's own Jenkins-Traub
root finder compiled from the reference tree (oracle/_ref/librpoly_ref.so) and against dense sampling (the
reference's own check, test_polynomial.cpp:81-135 / test_polynomial_optimization.cpp:690-727); the device lane
algorithm (mtg_extrema_lane.h) is run on the host (tests/extrema_emu.cpp) against the oracle.
GPU part: mtg_minmax_magnitude / mtg_scale_segment_times_to_meet_constraints through the C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import helpers
from oracle import oracle_extrema as ox
from oracle import oracle_np as onp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BACKEND = "ref" if ox.ref_available() else "numpy"


@pytest.fixture(scope="module")
def xemu():
    so = os.path.join(ROOT, "tests", "libmtg_extrema_emu.so")
    src = os.path.join(ROOT, "tests", "extrema_emu.cpp")
    hdr = os.path.join(ROOT, "mav_trajectory_generation_amd", "csrc", "mtg_extrema_lane.h")
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in (src, hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.extrema_emu_segments.argtypes = [ctypes.c_int] * 3 + [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p,
                                                               ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    lib.extrema_emu_roots22.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return lib


def solved_batch(n, k, dim, bsz, seed):
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, seed)
    coeffs, _, _ = onp.solve_batch(n, n // 2 - 1, masks, times, d_fixed)
    return masks, times, d_fixed, coeffs


@pytest.mark.skipif(not ox.ref_available(), reason="oracle/_ref/librpoly_ref.so not built (needs /root/reference)")
def test_numpy_roots_port_matches_reference_rpoly():
    """The numpy.roots port agrees with the reference's rpoly_ak1 on random polynomials of every degree the path
    produces (up to kMaxConvolutionSize = 22 coefficients), incl. trailing-zero stripping (rpoly_ak1.cpp:57-90)."""
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(1, 23))
        c = rng.uniform(-100.0, 100.0, n)
        if trial % 5 == 0:
            c[n - int(rng.integers(0, n)):] = 0.0      # trailing zeros (possibly all)
        ok_r, r_ref = ox.find_roots(c, "ref")
        ok_n, r_np = ox.find_roots(c, "numpy")
        assert len(r_ref) == len(r_np)
        if len(r_ref):
            a, b = np.sort_complex(r_ref), np.sort_complex(r_np)
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a))) < 1e-8


def test_oracle_extrema_vs_dense_sampling():
    """The reference's own acceptance test for the analytic extrema (test_polynomial_optimization.cpp:710-711:
    analytic vs sampled maximum, kTolerance) restated for the oracle."""
    _, times, _, coeffs = solved_batch(10, 5, 3, 6, 99)
    for b in range(coeffs.shape[0]):
        for der in (1, 2):
            _, mx, _ = ox.trajectory_min_max_magnitude(coeffs[b], times[b], der, None, BACKEND)
            sampled = 0.0
            for k in range(5):
                for t in np.linspace(0.0, times[b, k], 400):
                    sampled = max(sampled, np.sqrt(sum(ox.evaluate(coeffs[b, k, d], t, der) ** 2 for d in range(3))))
            assert mx[1] >= sampled - 1e-12
            assert mx[1] - sampled < 1e-3 * mx[1]


def test_lane_root_finder_vs_numpy(xemu):
    """Derivative-chain root isolation (mtg_extrema_lane.h real_roots_unit) finds exactly the real roots in [0, 1]."""
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(1500):
        deg = int(rng.integers(1, 22))
        g = np.zeros(22)
        if trial % 3 == 0:
            rts = rng.uniform(-0.5, 1.5, min(deg, 8))          # several real roots inside / outside the interval
            p = np.poly(rts)[::-1]
            g[:len(p)] = p
        else:
            g[:deg + 1] = rng.uniform(-1.0, 1.0, deg + 1)
        out = np.zeros(24)
        cnt = xemu.extrema_emu_roots22(g.ctypes.data, out.ctypes.data)
        _, r = ox.find_roots(g, "numpy")
        want = np.sort([x.real for x in r if abs(x.imag) < 1e-7 and 0.0 <= x.real <= 1.0])
        assert cnt == len(want), (trial, deg, want, out[:cnt])
        if cnt:
            assert np.all(np.diff(out[:cnt]) >= 0.0)
            worst = max(worst, float(np.max(np.abs(want - out[:cnt]))))
    assert worst < 1e-6     # clustered roots: numpy's companion eigenvalues are the less accurate side


def test_lane_root_finder_zero_and_constant_polynomials(xemu):
    out = np.zeros(24)
    g = np.zeros(22)
    assert xemu.extrema_emu_roots22(g.ctypes.data, out.ctypes.data) == 0      # all-zero: no roots (rpoly_ak1.cpp:75-79)
    g[0] = 3.0
    assert xemu.extrema_emu_roots22(g.ctypes.data, out.ctypes.data) == 0      # 0th order (rpoly_ak1.cpp:86-90)
    g[:] = 0.0
    g[0], g[1] = -0.25, 1.0
    assert xemu.extrema_emu_roots22(g.ctypes.data, out.ctypes.data) == 1 and abs(out[0] - 0.25) < 1e-15


@pytest.mark.parametrize("n,k,dim", [(10, 8, 3), (12, 4, 3), (8, 5, 3), (10, 3, 1), (10, 6, 4), (6, 3, 2), (4, 2, 2)])
def test_lane_segment_extrema_vs_oracle(xemu, n, k, dim):
    bsz = 12
    _, times, _, coeffs = solved_batch(n, k, dim, bsz, 300 + n + k)
    tt = np.ascontiguousarray(times)
    for der in (1, 2, 0):
        if n - der - 1 < 1:
            continue
        for dims in [list(range(dim))] + ([[dim - 1]] if dim > 1 else []):
            mask = sum(1 << d for d in dims)
            out = np.zeros((bsz, k, 4))
            assert xemu.extrema_emu_segments(n, k, dim, bsz, coeffs.ctypes.data, tt.ctypes.data, der, mask,
                                             out.ctypes.data) == 0
            for b in range(bsz):
                _, _, per = ox.trajectory_min_max_magnitude(coeffs[b], times[b], der, dims, BACKEND)
                scale = np.abs(per[:, 3]).max()
                helpers.assert_extrema_close(per, out[b], der, (der, dims, b))
                # the reported time reproduces the reported value
                for s in range(k):
                    v = np.sqrt(sum(ox.evaluate(coeffs[b, s, d], out[b, s, 2], der) ** 2 for d in dims))
                    assert abs(v - out[b, s, 3]) <= 1e-12 * scale


def test_oracle_scaling_meets_bounds_in_one_round():
    """scaleSegmentTimesToMeetConstraints: 'in vast majority of cases, this will converge within 1 iteration'
    (trajectory.cpp:388) -- stretching by s rescales the maxima exactly, so the second check passes."""
    _, times, _, coeffs = solved_batch(10, 6, 3, 5, 1234)
    for b in range(5):
        ok, segs, new_t, n_scaled = ox.scale_segment_times_to_meet_constraints(coeffs[b], times[b], 1.5, 2.0, BACKEND)
        assert ok and n_scaled <= 1
        v, a = ox.compute_max_velocity_and_acceleration(segs, new_t, BACKEND)
        assert v <= 1.5 * (1 + 1e-3) and a <= 2.0 * (1 + 1e-3)
        # shape unchanged: position at the end of every segment is the same
        for k in range(6):
            for d in range(3):
                assert abs(ox.evaluate(segs[k, d], new_t[k], 0) - ox.evaluate(coeffs[b, k, d], times[b, k], 0)) < 1e-9


def test_shared_root_search_is_bit_identical_on_the_host(xemu):
    """Round 4: two lanes may share one root search (each refines the brackets of its ranks).  A bracket's refinement does not
    depend on who refines it: the emulation of a shared search (one lane taking every part) returns the same bits."""
    _, times, _, coeffs = solved_batch(10, 8, 3, 12, 4711)
    xemu.extrema_emu_set_parts.argtypes = [ctypes.c_int]
    outs = []
    for parts in (1, 2, 3):
        xemu.extrema_emu_set_parts(parts)
        out = np.zeros((12, 8, 4))
        for der in (1, 2):
            assert xemu.extrema_emu_segments(10, 8, 3, 12, coeffs.ctypes.data, times.ctypes.data, der, 7, out.ctypes.data) == 0
            outs.append(out.copy())
    xemu.extrema_emu_set_parts(1)
    for k in range(2):
        assert np.array_equal(outs[k], outs[2 + k]) and np.array_equal(outs[k], outs[4 + k])
    # one code body for all levels (zero-padded full-length chains; a selectable variant) against the per-level bodies: the same
    # roots up to the rounding of the zero-padded Horner steps
    xemu.extrema_emu_set_rolled(1)
    out = np.zeros((12, 8, 4))
    assert xemu.extrema_emu_segments(10, 8, 3, 12, coeffs.ctypes.data, times.ctypes.data, 1, 7, out.ctypes.data) == 0
    xemu.extrema_emu_set_rolled(0)
    assert np.abs(out[..., 3] - outs[0][..., 3]).max() <= 1e-12 * np.abs(out[..., 3]).max()
    assert np.abs(out[..., 1] - outs[0][..., 1]).max() <= 1e-9 * np.abs(out[..., 3]).max()


# ----------------------------------------------------------------------------------------------- GPU (C ABI)
@pytest.fixture(scope="module")
def ctx():
    import mav_trajectory_generation_amd as m
    c = m.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,dim,bsz", [(10, 8, 3, 70), (12, 3, 4, 20), (8, 5, 1, 33), (6, 4, 2, 17)])
def test_gpu_minmax_magnitude_vs_oracle(ctx, n, k, dim, bsz):
    import torch
    import mav_trajectory_generation_amd as m
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 777 + n)
    plan = m.Plan(ctx, n, dim, k, n // 2 - 1, masks)
    t = torch.from_numpy(times).cuda()
    co, _, _ = plan.solve(t, torch.from_numpy(d_fixed).cuda())
    ctx.sync()
    host = co.cpu().numpy()
    for der, dims in [(1, None), (2, None), (1, [0]), (0, None)]:
        seg, traj, idx = m.minmax_magnitude(ctx, co, t, der, dims)
        seg_soa, traj_soa, idx_soa = m.minmax_magnitude(ctx, co, t.t().contiguous(), der, dims, times_layout="soa")
        ctx.sync()
        assert torch.equal(seg, seg_soa) and torch.equal(traj, traj_soa) and torch.equal(idx, idx_soa)
        seg, traj, idx = seg.cpu().numpy(), traj.cpu().numpy(), idx.cpu().numpy()
        dlist = list(range(dim)) if dims is None else dims
        for b in range(bsz):
            mn, mx, per = ox.trajectory_min_max_magnitude(host[b], times[b], der, dlist, BACKEND)
            scale = np.abs(per[:, 3]).max()
            helpers.assert_extrema_close(per, seg[b], der, (der, dims, b))
            assert abs(traj[b, 3] - mx[1]) <= (1e-9 if der else 1e-6) * scale and abs(traj[b, 1] - mn[1]) <= 1e-6 * scale
            # Extremum::segment_idx: the reported segment holds the reported value; equals the oracle's unless two
            # segments tie within round-off (shared vertices)
            assert seg[b, idx[b, 1], 3] == traj[b, 3] and seg[b, idx[b, 0], 1] == traj[b, 1]
            if idx[b, 1] != mx[2]:
                assert abs(per[idx[b, 1], 3] - mx[1]) <= 1e-6 * scale
    plan.close()


@pytest.mark.gpu
def test_gpu_scale_segment_times_vs_oracle(ctx):
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, bsz = 10, 6, 3, 40
    masks, times, d_fixed = helpers.reference_batch(bsz, k, n, dim, 4242)
    plan = m.Plan(ctx, n, dim, k, 4, masks)
    t = torch.from_numpy(times).cuda()
    co, _, _ = plan.solve(t, torch.from_numpy(d_fixed).cuda())
    ctx.sync()
    co0, t0 = co.cpu().numpy(), times.copy()
    v_max, a_max = 2.0, 2.5        # some trajectories already feasible (nfabian times were for v=3, a=5 but smooth)
    scaling, within, ws = m.scale_segment_times_to_meet_constraints(ctx, co, t, v_max, a_max)
    ctx.sync()
    got_c, got_t = co.cpu().numpy(), t.cpu().numpy()
    scaling, within = scaling.cpu().numpy(), within.cpu().numpy()
    n_scaled_total = 0
    for b in range(bsz):
        ok, segs, new_t, n_scaled = ox.scale_segment_times_to_meet_constraints(co0[b], t0[b], v_max, a_max, BACKEND)
        n_scaled_total += n_scaled
        assert bool(within[b]) == ok
        assert np.max(np.abs(got_t[b] - new_t) / new_t) <= 1e-9
        cs = np.abs(segs).max(axis=-1, keepdims=True)
        assert np.max(np.abs(got_c[b] - segs) / cs) <= 1e-9
        assert abs(scaling[b] - new_t.sum() / t0[b].sum()) <= 1e-9 * scaling[b]
    assert 0 < n_scaled_total       # the case exercises the scaling branch
    # SoA times + one round only: same scaled result (round 2 is only the verification)
    co2 = torch.from_numpy(co0).cuda()
    t2 = torch.from_numpy(t0).cuda().t().contiguous()
    _, within1, _ = m.scale_segment_times_to_meet_constraints(ctx, co2, t2, v_max, a_max, max_iterations=1,
                                                              times_layout="soa")
    ctx.sync()
    assert torch.equal(co2, co) and torch.equal(t2.t().contiguous(), t)
    assert int(within1.sum()) == int((scaling == 1.0).sum())    # 1 round: within_range is the pre-scaling check
    plan.close()


@pytest.mark.gpu
def test_gpu_extrema_full_size_properties(ctx):
    """At benchmark size: after scaling every trajectory meets the bounds, the analytic maxima dominate and match a
    dense sampling (the reference's own acceptance check), scaled trajectories keep their shape."""
    import torch
    import mav_trajectory_generation_amd as m
    from mav_trajectory_generation_amd.workload import ends_full_masks, random_waypoint_batch
    n, k, dim, bsz = 10, 8, 3, 100000
    masks = ends_full_masks(n, k)
    t, f = random_waypoint_batch(bsz, k, dim, n, masks, seed=5, device="cuda")
    plan = m.Plan(ctx, n, dim, k, 4, masks)
    co, _, _ = plan.solve(t, f)
    ctx.sync()
    total0 = t.sum(dim=1)
    end_pos0 = m.sample_range(ctx, co, t, 0.0, 1.0, 1, 1).clone()       # position at t = 0
    v_max, a_max = 2.0, 2.0
    scaling, within, ws = m.scale_segment_times_to_meet_constraints(ctx, co, t, v_max, a_max)
    ctx.sync()
    assert bool(within.all())
    assert bool((scaling >= 1.0).all()) and bool((scaling > 1.0).any())
    assert torch.allclose(t.sum(dim=1), total0 * scaling, rtol=1e-12)
    traj = ws[2 * bsz * k * 4:2 * bsz * k * 4 + 2 * bsz * 4].view(2, bsz, 4)
    assert bool((traj[0, :, 3] <= v_max * (1 + 1e-3)).all()) and bool((traj[1, :, 3] <= a_max * (1 + 1e-3)).all())
    # every scaled trajectory touches one of the two bounds; unscaled ones are below both
    tight = torch.maximum(traj[0, :, 3] / v_max, torch.sqrt(traj[1, :, 3] / a_max))
    assert bool((tight[scaling > 1.0] > 1 - 1e-9).all()) and bool((tight <= 1 + 1e-9).all())
    # dense sampling of a subset (test_polynomial_optimization.cpp:710-711 style)
    sub = slice(0, 2000)
    S = 2001
    dt = float(t[sub].sum(dim=1).max()) / (S - 1)
    smp, nv = m.sample_range(ctx, co[sub].contiguous(), t[sub].contiguous(), 0.0, dt, S, 3, want_valid=True)
    ctx.sync()
    vel = smp[:, :, 1, :].norm(dim=-1).max(dim=1).values
    acc = smp[:, :, 2, :].norm(dim=-1).max(dim=1).values
    assert bool((vel <= traj[0, sub, 3] * (1 + 1e-12)).all()) and bool((acc <= traj[1, sub, 3] * (1 + 1e-12)).all())
    assert bool((vel >= traj[0, sub, 3] * (1 - 2e-3)).all()) and bool((acc >= traj[1, sub, 3] * (1 - 2e-3)).all())
    assert torch.allclose(m.sample_range(ctx, co, t, 0.0, 1.0, 1, 1), end_pos0, rtol=0, atol=1e-12)
    plan.close()


@pytest.mark.gpu
def test_gpu_extrema_argument_errors(ctx):
    import torch
    import mav_trajectory_generation_amd as m
    co = torch.zeros((2, 3, 2, 10), dtype=torch.float64, device="cuda")
    t = torch.ones((2, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(m.MtgError):
        m.minmax_magnitude(ctx, co, t, 10)            # N - derivative - 1 < 0 (polynomial.cpp:70-73)
    with pytest.raises(m.MtgError):
        m.minmax_magnitude(ctx, co, t, 1, [2])        # dimension out of bounds (segment.cpp:102-107)
    with pytest.raises(m.MtgError):
        m.scale_segment_times_to_meet_constraints(ctx, co, t, 0.0, 1.0)
    seg, traj, idx = m.minmax_magnitude(ctx, co, t, 1)        # all-zero polynomials: no roots, extrema 0 at t_start
    ctx.sync()
    assert float(seg.abs().max()) == 0.0 and int(idx.abs().max()) == 0


@pytest.mark.gpu
def test_gpu_shared_root_search_is_bit_identical(ctx):
    """One, two and four lanes per root search (the default picks by launch size): the same bits, for the extrema tables and for
    the time scaling built on them."""
    import torch
    import mav_trajectory_generation_amd as m
    n, k, dim, bsz = 10, 8, 3, 3000
    masks = m.ends_full_masks(n, k)
    t, f = m.random_waypoint_batch(bsz, k, dim, n, masks, seed=12, device="cuda")
    plan = m.Plan(ctx, n, dim, k, 4, masks)
    co, _, _ = plan.solve(t, f)
    ctx.sync()
    res = {}
    try:
        for split in (1, 2, 3, 0):
            ctx.set_option("extrema_split", split)      # (bits 0-1: lanes per search: 1, 2, 3 = four; 0 = by launch size)
            seg_v, traj_v, idx_v = m.minmax_magnitude(ctx, co, t, 1)
            seg_a, traj_a, _ = m.minmax_magnitude(ctx, co, t, 2, dimensions=[0, 2])
            c2, t2 = co.clone(), t.clone()
            sc, within, _ = m.scale_segment_times_to_meet_constraints(ctx, c2, t2, 1.5, 2.0)
            ctx.sync()
            res[split] = (seg_v, traj_v, idx_v, seg_a, traj_a, c2, t2, sc, within)
    finally:
        ctx.set_option("extrema_split", -1)
    for other in (2, 3, 0):
        for a, b in zip(res[1], res[other]):
            assert torch.equal(a, b), other
    plan.close()
